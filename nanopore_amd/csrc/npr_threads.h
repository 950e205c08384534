// npr_threads.h -- host worker threads shared by the C ABI's host stages (npr_api.cpp, npr_io.cpp).  Not installed.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

namespace npr {

// CPUs this process may actually use: the hardware count capped by the cgroup CPU quota (a container on a 256-core
// host is often limited to a few cores; running 256 threads inside such a quota is slower than running 16)
inline int granted_cpus() {  // (read once per process: the quota file does not change under a running job)
    static const int granted = [] {
        int n = static_cast<int>(std::max(1u, std::thread::hardware_concurrency()));
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char quota[32] = {0};
            long long period = 0;
            if (std::fscanf(f, "%31s %lld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
                const long long q = std::atoll(quota);
                if (q > 0) n = std::min<int>(n, static_cast<int>(std::max<long long>(1, (q + period - 1) / period)));
            }
            std::fclose(f);
        }
        return n;
    }();
    return granted;
}
inline int usable_cpus() {
    if (const char *e = std::getenv("NPR_HOST_THREADS")) return std::max(1, std::atoi(e));  // (a caller may change it between calls: bench.py does)
    return granted_cpus();
}

// Runs f(0..n-1) on `threads` host threads.  An exception inside a worker (std::bad_alloc from a plan or MEA vector)
// must not escape the thread -- that would be std::terminate --: it is caught, the remaining items are skipped and the
// first one is rethrown on the calling thread, where the C ABI turns it into NPR_ERR_NOMEM.
template <typename F>
void parallel_for(int64_t n, int threads, F f) {
    threads = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(threads, n)));
    if (threads == 1) {
        for (int64_t i = 0; i < n; ++i) f(i);
        return;
    }
    std::atomic<int64_t> next{0};
    std::atomic<bool> failed{false};
    std::exception_ptr first;
    std::mutex mu;
    std::vector<std::thread> pool;
    auto work = [&] {
        try {
            for (;;) {
                const int64_t i = next.fetch_add(1);
                if (i >= n || failed.load()) break;
                f(i);
            }
        } catch (...) {
            std::lock_guard<std::mutex> lock(mu);
            if (!first) first = std::current_exception();
            failed = true;
        }
    };
    pool.reserve(static_cast<size_t>(threads));
    for (int t = 0; t + 1 < threads; ++t) {
        try {
            pool.emplace_back(work);
        } catch (const std::system_error &) {  // the process's thread limit: the threads that did start (and this one) share the items
            break;
        }
    }
    work();  // the calling thread is one of the workers
    for (auto &th : pool) th.join();
    if (first) std::rethrow_exception(first);
}

}  // namespace npr
