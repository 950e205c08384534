// npr_api.cpp -- the C ABI of libnprealign (include/nprealign.h): context, model slots, batch staging,
// launch of the DP kernels, result gathering.  Replaces the per-read process fan-out / temp-file gather
// of nanopore/analyses/utils.py:557-609 by one batched call.  There is no CPU execution path for the DP:
// without a usable gfx950 device npr_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "npr_device.h"
#include "npr_internal.h"
#include "npr_sched.h"
#include "npr_threads.h"

using namespace npr;

struct npr_plan {
    Plan plan;
};

namespace {
struct MeaScratch;

// Forward-value scratch of ONE device (one region per resident wavefront), shared by every context on that device and only
// growing: a hipMalloc of ~100 GB costs seconds, far more than the DP pass it serves, and a pipelined job keeps two
// batches in flight on two contexts of the same GPU (nanopore_amd/job.py) -- their DP launches each fill the chip and so
// run one after the other anyway, and one arena instead of two is the difference between fitting the device and not
// (config 3: ~130-250 GB).  `mu` is held by whatever launches kernels that read or write the arena (the DP pass, the device
// MEA stage whose tables are carved out of it, the E-step, the dense dump) until they have finished, and while it is
// regrown.  `epoch` is bumped whenever its contents may have been overwritten: a finished batch may use the packed cigars
// the MEA stage left there only while its stamp is current.
// The arena points kArenaPad bytes into its allocation and is followed by as much: the register E-step loads forward rows
// with a slot shift of up to two and may touch a few cells before / after a region.
struct DeviceArena {
    static constexpr size_t kPad = 1024;
    std::mutex mu;
    char *F = nullptr;  // 8 bytes per cell
    std::atomic<size_t> cells{0};  // (read without the mutex where only its size matters: staging must not wait for a DP pass)
    std::atomic<uint64_t> epoch{1};
    int users = 0;
};
constexpr int kMaxDevices = 64;
DeviceArena g_arena[kMaxDevices];
}  // namespace

struct npr_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // one side stream per kernel class so that the classes of a mixed batch run concurrently instead of each
    // leaving the chip idle during its tail
    static constexpr int kSideStreams = 6;
    hipStream_t side[kSideStreams] = {};
    hipEvent_t side_done[kSideStreams] = {};
    int cu_count = 0;
    size_t total_mem = 0;
    bool model_set[NPR_MAX_MODELS] = {};
    DevModel models[NPR_MAX_MODELS];
    DevModel *d_models = nullptr;
    std::string last_error;
    int host_threads = 1;
    DeviceArena *arena = nullptr;  // the device's forward scratch (shared with the other contexts on this device)
    int overlap = 0;               // NPR_OPT_OVERLAP: see include/nprealign.h (1: own MEA tables + half of every SIMD left free by the DP launches; 2: own MEA tables only)
    int64_t opt[NPR_OPT_COUNT] = {};  // npr_ctx_option: the test / bring-up switches (all 0 by default)
    static constexpr size_t kArenaPad = DeviceArena::kPad;
    float *arena_Fx = nullptr;  // E-step only: four more forward planes (per context)
    size_t arena_fx_cells = 0;
    // pinned host staging for the posterior triples of npr_batch_finish (grow-only): a pageable destination halves
    // the D2H rate and the copy is a GB per batch
    void *pin_pairs = nullptr;
    size_t pin_pairs_bytes = 0;
    std::vector<hipEvent_t> ops_events;  // one per piece of the ops' D2H (device_mea)
    // the packed cigars of the last batch or two that were destroyed: a batch's 75-150 MB, whose pages cost 3 ms to touch when the
    // next batch is finished and 6 ms to give back when it is destroyed (with a caller waiting for the context)
    struct HostWords {
        std::unique_ptr<uint32_t[]> p;
        int64_t cap = 0;
    };
    std::vector<HostWords> packed_pool;
    // pinned host staging of npr_batch_create (plan points + sequence windows), grow-only
    void *pin_stage = nullptr;
    size_t pin_stage_bytes = 0;
    MeaScratch *mea = nullptr;
    // Device buffers of destroyed batches, kept for the next batch (DevBuf::alloc_from): hipMalloc / hipFree of the
    // gigabyte-sized band, control-word and pair arrays cost more than the kernels that fill them (0.1 s per batch of
    // 50 k reads), and a pipeline stages batch after batch of the same shape.
    struct Cached {
        void *p;
        size_t bytes;
    };
    std::vector<Cached> cache;
    size_t cache_bytes = 0;
    void cache_flush() {
        for (const Cached &c : cache) (void)hipFree(c.p);
        cache.clear();
        cache_bytes = 0;
    }
};

namespace {

// NPR_POISON=<byte>: every device buffer is filled with that byte when it is handed out (and the forward scratch before
// every batch), so that a kernel reading memory nobody wrote gives the same wrong answer on every box instead of
// depending on what the previous owner of the memory left there.  Test / bring-up switch.
int poison_byte() {
    const char *e = std::getenv("NPR_POISON");
    return e && e[0] ? static_cast<int>(std::strtol(e, nullptr, 0)) & 0xff : -1;
}
void poison(void *p, size_t bytes) {
    if (poison_byte() >= 0 && p && bytes) {
        (void)hipMemset(p, poison_byte(), bytes);
        (void)hipDeviceSynchronize();
    }
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t count = 0, cap = 0;
    hipError_t alloc(size_t n) {
        release();
        count = n;
        if (n == 0) return hipSuccess;
        const hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T));
        if (e == hipSuccess) poison(p, n * sizeof(T));
        return e;
    }
    void release() {
        if (p && !borrowed) {
            if (owner && owner->cache.size() < 160) {
                owner->cache.push_back(npr_ctx::Cached{p, held});
                owner->cache_bytes += held;
            } else {
                (void)hipFree(p);
            }
        }
        p = nullptr;
        count = 0, cap = 0, borrowed = false, owner = nullptr, held = 0;
    }
    // a view of memory owned elsewhere (the forward scratch arena, idle between the DP launch and the next one)
    bool borrowed = false;
    void borrow(T *ptr, size_t n) {
        release();
        p = ptr, count = n, borrowed = true;
    }
    size_t bytes() const { return count * sizeof(T); }
    // a buffer from the context's cache of released ones (the smallest that fits without wasting more than half), else a
    // fresh one; it goes back to the cache when released
    npr_ctx *owner = nullptr;
    size_t held = 0;
    hipError_t alloc_from(npr_ctx *ctx, size_t n) {
        release();
        count = n;
        if (n == 0) return hipSuccess;
        const size_t need = n * sizeof(T);
        // small ones come in 256 KiB pieces and any cached piece up to 1 MiB serves them: a batch makes a dozen tables of a few
        // words per task, and hipFree of each (synchronous) cost 2-3 ms when the batch was staged
        constexpr size_t kSmall = size_t(1) << 20, kPiece = size_t(256) << 10;
        int best = -1;
        for (size_t i = 0; i < ctx->cache.size(); ++i)
            if (ctx->cache[i].bytes >= need && (ctx->cache[i].bytes <= 2 * need || ctx->cache[i].bytes <= kSmall) &&
                (best < 0 || ctx->cache[i].bytes < ctx->cache[best].bytes))
                best = static_cast<int>(i);
        if (best >= 0) {
            p = static_cast<T *>(ctx->cache[best].p), held = ctx->cache[best].bytes, owner = ctx;
            ctx->cache_bytes -= held;
            ctx->cache.erase(ctx->cache.begin() + best);
            poison(p, held);
            return hipSuccess;
        }
        const size_t take = need < kSmall ? (need + kPiece - 1) / kPiece * kPiece : need + need / 8;  // a little headroom: the next batch of the same shape differs by a few percent
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), take);
        if (e != hipSuccess && !ctx->cache.empty()) {
            (void)hipGetLastError();
            ctx->cache_flush();
            e = hipMalloc(reinterpret_cast<void **>(&p), take);
        }
        if (e == hipSuccess) held = take, owner = ctx, poison(p, take);
        return e;
    }
    // grow-only use (scratch kept from batch to batch): count is the size asked for, cap what is allocated
    hipError_t reserve(size_t n) {
        if (n <= cap && p) {
            count = n;
            return hipSuccess;
        }
        const hipError_t e = alloc(n + n / 4 + 1);
        cap = e == hipSuccess ? count : 0;
        count = e == hipSuccess ? n : 0;
        return e;
    }
    ~DevBuf() { release(); }
};

// scratch of the device MEA stage (npr_mea.hip), kept by the context: hipMalloc / hipFree of gigabytes per batch
// cost more than the kernels
struct MeaScratch {
    DevBuf<int64_t> off, mass, od;
    DevBuf<int32_t> cnt, start, col, sorted, small, tmp, map, pieces;
    DevBuf<uint32_t> dense;
};

int32_t fail(npr_ctx *ctx, int32_t code, const char *what, hipError_t e = hipSuccess) {
    // a launch or copy that finds the device full (a kernel's private segment is allocated at launch) is the same condition as
    // a failed hipMalloc: callers halve the batch and try again on NPR_ERR_NOMEM
    if (code == NPR_ERR_HIP && e == hipErrorOutOfMemory) code = NPR_ERR_NOMEM, (void)hipGetLastError();
    if (ctx) {
        ctx->last_error = what;
        if (e != hipSuccess) {
            ctx->last_error += ": ";
            ctx->last_error += hipGetErrorString(e);
        }
    }
    return code;
}

#define HIP_TRY(ctx, expr)                                                  \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) return fail((ctx), NPR_ERR_HIP, #expr, _e);   \
    } while (0)

// NPR_TIMING=1 prints host-stage wall times to stderr (bring-up / DESIGN.md host-inclusive numbers)
struct StageTimer {
    bool on;
    std::chrono::steady_clock::time_point t0;
    const char *what;
    explicit StageTimer(const char *w) : on(std::getenv("NPR_TIMING") != nullptr), t0(std::chrono::steady_clock::now()), what(w) {}
    void lap(const char *label) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[npr timing] %s / %s: %.1f ms\n", what, label, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

}  // namespace

struct npr_batch {
    npr_ctx *ctx = nullptr;
    npr_params params{};
    int64_t n_reads = 0;
    // host copies needed by finish()
    std::vector<int64_t> ref_len, read_len;  // spans of the guide's window
    std::vector<int64_t> gstart;             // per read: first reference / read position of the window
    std::vector<int32_t> ref_id;             // per read: its reference sequence
    std::vector<int32_t> guide_ops;
    std::vector<int64_t> guide_off;
    std::vector<int32_t> read_status;    // planning status per read
    std::vector<int32_t> read_first_task, read_ntasks;
    std::vector<Task> tasks;             // device order (sorted longest first)
    std::vector<int32_t> task_of;        // [read_first_task[r] + s] -> index into tasks
    std::vector<int64_t> task_cells;     // in-band lattice cells per task (device order)
    std::vector<TaskOut> outs;
    std::vector<uint8_t> task_rerun;     // row-scaled tasks npr_batch_run ran again with a per-cell exponent
    npr_batch_stats stats{};
    // device
    DevBuf<Task> d_tasks;
    DevBuf<TaskOut> d_outs;
    DevBuf<int32_t> d_queue;
    DevBuf<uint8_t> d_seq;
    DevBuf<int32_t> d_lo, d_n;
    DevBuf<uint32_t> d_coff;
    DevBuf<uint32_t> d_ctl;  // register-kernel tasks: frame schedule, two words per anti-diagonal
    DevBuf<Stripe> d_stripes;  // k_dp_tile tasks: stripe tables
    DevBuf<uint32_t> d_rowmask;  // ... and the packed lane masks of every row of every stripe (tile_row_word)
    DevBuf<PlanSeg> d_pseg;    // the segments as the device planner sees them (read order)
    DevBuf<int64_t> d_region;  // k_dp_tile: first scratch cell of each resident workgroup
    std::vector<int64_t> region_end;  // ... and one past its last (host copy: the E-step sizes its planes for the regions it uses)
    size_t scratch_cells = 0;  // forward scratch this batch needs from the context arena
    bool variable_regions = false;  // the one-wavefront frame launches have regions of their own size (not E-step capable)
    bool pair_rs = false;  // the batch was staged for the row-scaled kernels (classes 12-17: k_dp_mid_rs, k_dp_rs)
    DevBuf<int32_t> d_px, d_py;
    DevBuf<float> d_pp;
    int64_t slot_stride = 0;
    // One DP launch per kernel class present in the batch (tasks are grouped by class, longest first).
    struct Launch {
        int cls;      // index into kClassTab; (historical note) 0..2 register staircase kernel with 1/2/4 cells per lane; 3..5 generic kernel with an LDS ring for
                      // bands of at most 512 / 1024 / 2270 cells; 6 generic kernel with the ring in HBM/L2
        int first, count, grid, wcap;
        int threads;  // generic kernel: workgroup size (wavefronts per task x 64)
        size_t lds;
        int64_t cells;
        int64_t width;  // widest anti-diagonal of the class
        int slot_base;  // first forward-scratch region of this launch
        int region_first;  // own_regions: index of its first entry in d_region
        bool own_regions;  // one region per workgroup sized by its first task (d_region) instead of uniform ones
    };
    std::vector<Launch> launches;
    DevBuf<float> d_ring;
    bool ran = false, finished = false;
    // results
    std::vector<npr_read_result> results;
    std::vector<int64_t> ops_off;
    std::unique_ptr<int32_t[]> ops;      // (op, length) pairs of all reads; not a vector: no zero-fill of 100s of MB
    int64_t ops_words = 0, ops_cap = 0;
    // the same cigars as one 32-bit word per op (length << 2 | op): how the device MEA stage hands them over.  Either
    // form is made from the other the first time it is asked for.
    std::unique_ptr<uint32_t[]> packed;
    int64_t packed_cap = 0;
    bool have_pairs_form = false, have_packed_form = false;
    // NPR_MODE_RESCORE_ORIGINAL: the cigars are the guide's (operations of length 0 left out), made from b->guide_ops the first time somebody asks
    bool ops_from_guide = false;
    // ... and what npr_batch_create leaves for npr_batch_finish: the guide's M columns as a table on the device (rescore_stage), per read the
    // number of M columns and of operations kept; rs_staged = false: the host stage scores (NPR_OPT_HOST_MEA, or a sum that could not be exact)
    bool rs_staged = false;
    int rs_shift = 0;
    std::vector<int64_t> rs_columns, rs_kept;
    DevBuf<int32_t> d_rs_gy;
    DevBuf<int64_t> d_rs_gx_off;
    std::vector<int64_t> pair_off;
    std::vector<Pair> pairs;             // filled by fetch_pairs(): at finish in the host modes, on demand after the device MEA
    bool pairs_ready = false;
    std::vector<int64_t> task_dst;       // prefix of the per-task pair counts
    // packed cigars left on the device by the device MEA stage (valid while dev_ops_epoch == the arena's epoch)
    const uint32_t *dev_ops = nullptr;
    const int64_t *dev_od = nullptr;
    uint64_t dev_ops_epoch = 0;
};

extern "C" {

int32_t npr_abi_version(void) { return NPR_ABI_VERSION; }

const char *npr_strerror(int32_t code) {
    switch (code) {
        case NPR_OK: return "ok";
        case NPR_ERR_INVALID: return "invalid argument or guide alignment not global";
        case NPR_ERR_ZERO_PROB: return "total probability is zero inside the band";
        case NPR_ERR_CAPACITY: return "output capacity exceeded";
        case NPR_ERR_MODEL: return "unsupported or malformed HMM";
        case NPR_ERR_NO_DEVICE: return "no usable gfx950 device (there is no CPU fallback)";
        case NPR_ERR_HIP: return "HIP runtime error";
        case NPR_ERR_BAND_TOO_WIDE: return "band wider than the kernels support";
        case NPR_ERR_NOMEM: return "out of memory";
        case NPR_ERR_STATE: return "call sequence violated";
        default: return "unknown error";
    }
}

int32_t npr_create(int32_t device_id, npr_ctx **out, char *err, size_t errlen) {
    auto say = [&](const char *msg, hipError_t e) {
        if (err && errlen) std::snprintf(err, errlen, "%s%s%s", msg, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
    };
    if (!out || device_id < 0) {
        say("npr_create: bad arguments", hipSuccess);
        return NPR_ERR_INVALID;
    }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        say("npr_create: no HIP device visible", e);
        return NPR_ERR_NO_DEVICE;
    }
    if (device_id >= count || device_id >= kMaxDevices) {
        say("npr_create: device index out of range", hipSuccess);
        return NPR_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) {
        say("npr_create: hipGetDeviceProperties", e);
        return NPR_ERR_NO_DEVICE;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        if (err && errlen) std::snprintf(err, errlen, "npr_create: device %d is %s, this library is built for gfx950 only", device_id, prop.gcnArchName);
        return NPR_ERR_NO_DEVICE;
    }
    e = hipSetDevice(device_id);
    if (e != hipSuccess) {
        say("npr_create: hipSetDevice", e);
        return NPR_ERR_NO_DEVICE;
    }
    // every failure below goes through npr_destroy (streams, events and device memory made so far are released) and the
    // handle is only handed out once the default model is installed
    npr_ctx *ctx = new (std::nothrow) npr_ctx;
    if (!ctx) return NPR_ERR_NOMEM;
    ctx->device = device_id;
    ctx->arena = &g_arena[device_id];
    {
        std::lock_guard<std::mutex> lock(ctx->arena->mu);
        ++ctx->arena->users;
    }
    ctx->cu_count = prop.multiProcessorCount;
    ctx->total_mem = prop.totalGlobalMem;
    ctx->host_threads = usable_cpus();
    for (int i = 0; i < npr_ctx::kSideStreams; ++i)
        if ((e = hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->side_done[i], hipEventDisableTiming)) != hipSuccess) {
            say("npr_create: side stream allocation", e);
            npr_destroy(ctx);
            return NPR_ERR_HIP;
        }
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess ||
        (e = hipMalloc(reinterpret_cast<void **>(&ctx->d_models), sizeof(DevModel) * NPR_MAX_MODELS)) != hipSuccess) {
        say("npr_create: stream/event/model allocation", e);
        npr_destroy(ctx);
        return NPR_ERR_HIP;
    }
    // slot 0 defaults to the stock model (no --loadHmm)
    const int32_t rc = npr_set_hmm(ctx, 0, nullptr, nullptr);
    if (rc != NPR_OK) {
        if (err && errlen) std::snprintf(err, errlen, "npr_create: %s", ctx->last_error.c_str());
        npr_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return NPR_OK;
}

void npr_destroy(npr_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    ctx->cache_flush();
    if (ctx->d_models) (void)hipFree(ctx->d_models);
    if (ctx->arena) {  // the last context on the device takes the shared scratch with it
        std::lock_guard<std::mutex> lock(ctx->arena->mu);
        if (--ctx->arena->users == 0) {
            if (ctx->arena->F) (void)hipFree(ctx->arena->F - DeviceArena::kPad);
            ctx->arena->F = nullptr, ctx->arena->cells = 0, ++ctx->arena->epoch;
        }
    }
    if (ctx->arena_Fx) (void)hipFree(reinterpret_cast<char *>(ctx->arena_Fx) - npr_ctx::kArenaPad);
    if (ctx->pin_pairs) (void)hipHostFree(ctx->pin_pairs);
    if (ctx->pin_stage) (void)hipHostFree(ctx->pin_stage);
    delete ctx->mea;
    for (hipEvent_t ev : ctx->ops_events) (void)hipEventDestroy(ev);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    for (int i = 0; i < npr_ctx::kSideStreams; ++i) {
        if (ctx->side[i]) (void)hipStreamDestroy(ctx->side[i]);
        if (ctx->side_done[i]) (void)hipEventDestroy(ctx->side_done[i]);
    }
    delete ctx;
}

const char *npr_last_error(npr_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

// NPR_OPT_RELEASE_SCRATCH: the device's forward scratch (shared by the contexts of the device, regrown by the next batch that needs
// it) and this context's cache of released device buffers go back to the driver -- a process that is done with a big batch
// and stays alive (a pipeline's parent, a test session) need not keep a hundred GB of HBM from the next one.
static int32_t release_scratch(npr_ctx *ctx, bool caches_only) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->cache_flush();
    if (caches_only) return NPR_OK;  // (value 2: what a pipeline does when a batch does not fit, before it halves it: the scratch may be in use)
    if (ctx->arena) {
        std::lock_guard<std::mutex> lock(ctx->arena->mu);
        if (ctx->arena->F) (void)hipFree(ctx->arena->F - DeviceArena::kPad);
        ctx->arena->F = nullptr, ctx->arena->cells = 0, ++ctx->arena->epoch;
    }
    if (ctx->arena_Fx) (void)hipFree(reinterpret_cast<char *>(ctx->arena_Fx) - npr_ctx::kArenaPad);
    ctx->arena_Fx = nullptr, ctx->arena_fx_cells = 0;
    delete ctx->mea;
    ctx->mea = nullptr;
    ctx->packed_pool.clear();
    return NPR_OK;
}

int32_t npr_ctx_option(npr_ctx *ctx, int32_t option, int64_t value) {
    if (!ctx) return NPR_ERR_INVALID;
    switch (option) {
        case NPR_OPT_OVERLAP: ctx->overlap = value == 2 ? 2 : (value != 0 ? 1 : 0); return NPR_OK;
        case NPR_OPT_RELEASE_SCRATCH: return release_scratch(ctx, value == 2);
        default:
            if (option > NPR_OPT_RELEASE_SCRATCH && option < NPR_OPT_COUNT) {
                ctx->opt[option] = value;
                return NPR_OK;
            }
            return fail(ctx, NPR_ERR_INVALID, "npr_ctx_option: unknown option");
    }
}

int32_t npr_set_hmm(npr_ctx *ctx, int32_t slot, const double *T25, const double *E80) {
    if (!ctx || slot < 0 || slot >= NPR_MAX_MODELS) return NPR_ERR_INVALID;
    double T[25], E[80];
    if (T25 && E80) {
        std::memcpy(T, T25, sizeof(T));
        std::memcpy(E, E80, sizeof(E));
    } else if (!T25 && !E80) {
        stock_model(T, E);
    } else {
        return fail(ctx, NPR_ERR_INVALID, "npr_set_hmm: T and E must both be given or both be NULL");
    }
    DevModel m;
    const int32_t rc = make_dev_model(T, E, m);
    if (rc != NPR_OK) return fail(ctx, rc, "npr_set_hmm: model has a transition outside the five-state cell update, or a negative / non-finite entry");
    ctx->models[slot] = m;
    ctx->model_set[slot] = true;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(ctx->d_models + slot, &m, sizeof(DevModel), hipMemcpyHostToDevice));
    return NPR_OK;
}

// --------------------------------------------------------------------------------------------------
// Frame schedule of the register kernel (npr_kernel_stair.hip).  The wavefront holds a frame of C = 64*R slots of the
// current anti-diagonal (times NW wavefronts for k_dp_wide), slot j = lattice point (x0 + j, y0 - j); the frame takes an X-step (x0 += 1) into every odd
// anti-diagonal and a Y-step (y0 += 1) into every even one, so its first x-y, flo, just alternates.  The band (first
// x-y `lo`, n cells) must stay inside the frame; when it drifts to an edge the frame is REBASED by one slot
// (flo +- 2) between two anti-diagonals.  A rebase towards higher x-y may only precede an X-step and one towards
// lower x-y a Y-step (the kernel re-injects the base that left the wavefront at the step before), so the decision
// looks one anti-diagonal ahead.  Control words per anti-diagonal: row offset in the forward scratch (cells), and
// jlo | n << 13 | (rebase + 1) << 26.  Returns false when the band cannot be followed; `ctl` and `cells` may be null.
// --------------------------------------------------------------------------------------------------
namespace {

bool build_stair_schedule(const Segment &s, int R, int NW, uint32_t *ctl, int64_t *cells) {
    if (s.n.empty()) return false;
    return stair_schedule(s.lo.data(), s.n.data(), s.D(), s.max_width, R, NW, ctl, cells);
}

}  // namespace

// Kernel classes of a batch, each launched on its own: the register kernel with one wavefront per task (R slots per
// lane), the register kernel with NW wavefronts per task (k_dp_wide), the generic kernel with an LDS ring in three
// width classes, the generic kernel with its ring in HBM.
namespace {
enum { K_STAIR = 0, K_WIDE = 1, K_GENERIC_LDS = 2, K_GENERIC_GLOBAL = 3, K_TILE = 4, K_MID = 5, K_RS = 6, K_TILE_RS = 7 };
struct KClass {
    int kind, R, NW;
    int slots() const { return 64 * R * NW; }
};
constexpr int kClasses = 19;
constexpr KClass kClassTab[kClasses] = {{K_STAIR, 1, 1}, {K_STAIR, 2, 1}, {K_STAIR, 4, 1}, {K_WIDE, 2, 4}, {K_WIDE, 2, 8},
                                        {K_WIDE, 4, 8}, {K_WIDE, 4, 12}, {K_GENERIC_LDS, 0, 0}, {K_GENERIC_LDS, 0, 0},
                                        {K_GENERIC_LDS, 0, 0}, {K_GENERIC_GLOBAL, 0, 0}, {K_TILE, 2, 0},
                                        // k_dp_mid_rs<R>: the one-wavefront frame classes in row-scaled arithmetic with the two sweeps on two wavefronts that meet in the middle
                                        {K_MID, 1, 1}, {K_MID, 2, 1}, {K_MID, 4, 1},
                                        // k_dp_rs<R>: the one-wavefront frame classes 0-2 in row-scaled arithmetic (npr_rs.h)
                                        {K_RS, 1, 1}, {K_RS, 2, 1}, {K_RS, 4, 1},
                                        // k_dp_tile_cs: class 11's column stripes in column-scaled arithmetic (one exponent per lane of a stripe)
                                        {K_TILE_RS, 2, 0}};
constexpr int kFirstGeneric = 7, kTileClass = 11, kFirstPair = 12, kFirstRs = 15, kTileRsClass = 18, kQueueSlots = 24;
inline bool is_register_class(int c) { return kClassTab[c].kind <= K_WIDE || kClassTab[c].kind == K_MID || kClassTab[c].kind == K_RS; }
inline bool is_one_wave_kind(int kind) { return kind == K_STAIR || kind == K_RS; }
inline bool is_tile_kind(int kind) { return kind == K_TILE || kind == K_TILE_RS; }  // column stripes, NW wavefronts per task  // one wavefront per task on the frame schedule
// resident wavefronts per CU of the one-wavefront frame kernels (VGPR-limited: 71 / 80 / 162 registers: 7 / 6 / 3 per SIMD)
inline int stair_waves_per_cu(int R) { return R == 1 ? 28 : (R == 2 ? 24 : 12); }
// ... and of k_dp_rs<R> (72 / 72 / 105 registers: 7 / 7 / 4 per SIMD; R = 2 measured at 6 / 7 / 8 per SIMD in round 4: 7 is best)
inline int rs_waves_per_cu(int R) { return R == 1 ? 28 : (R == 2 ? 28 : 16); }

// Whether the row-scaled arithmetic (npr_rs.h) may be used with a model: its rows are renormalised to 2^NPR_RS_TOP every
// NPR_RS_K anti-diagonals with 2^6 of headroom, so nothing may grow by more than 2^(6 / NPR_RS_K) per anti-diagonal -- the sum of
// the transitions into a state times that state's largest emission (0.57 for the shipped models: values only shrink).
bool rs_model_ok(const DevModel &m) {
    double grow = 0.0, em_max = 0.0;
    for (int x = 0; x < 4; ++x)
        for (int y = 0; y < 4; ++y) em_max = std::max(em_max, static_cast<double>(m.em[x * 5 + y]));
    for (int to = 0; to < 5; ++to) {
        double col = 0.0, e = em_max;
        for (int from = 0; from < 5; ++from) col += static_cast<double>(m.T[from * 5 + to]);
        if (to > 0) {
            e = 0.0;
            for (int b2 = 0; b2 < 5; ++b2) e = std::max(e, static_cast<double>((to == 1 || to == 3) ? m.ex[to * 5 + b2] : m.ey[to * 5 + b2]));
        }
        grow = std::max(grow, col * e);
    }
    // ... and the backward sweep grows by the ROW sums: a state's value is the sum over its successors of transition times the
    // successor's emission (a stochastic model's rows sum to 1; a user's model need not be stochastic)
    auto emax = [&](int st) {
        if (st == 0) return em_max;
        double e = 0.0;
        for (int b2 = 0; b2 < 5; ++b2) e = std::max(e, static_cast<double>((st == 1 || st == 3) ? m.ex[st * 5 + b2] : m.ey[st * 5 + b2]));
        return e;
    };
    for (int from = 0; from < 5; ++from) {
        double row = 0.0;
        for (int to = 0; to < 5; ++to) row += static_cast<double>(m.T[from * 5 + to]) * emax(to);
        grow = std::max(grow, row);
    }
    return grow <= std::exp2(6.0 / NPR_RS_K);
}

// Whether every loaded model emits every base from every gap state with probability exactly 2^-2 (N included: make_dev_model gives it 1/4): the
// row-scaled kernels then take the gap emissions from a select instead of their LDS tables (npr_rs.h rs_cell_emissions; same bits).
bool flat_gap_emissions(const npr_ctx *ctx) {
    for (int sl = 0; sl < NPR_MAX_MODELS; ++sl) {
        if (!ctx->model_set[sl]) continue;
        const DevModel &m = ctx->models[sl];
        for (int b2 = 0; b2 < 5; ++b2)
            if (m.ex[5 + b2] != 0.25f || m.ex[15 + b2] != 0.25f || m.ey[10 + b2] != 0.25f || m.ey[20 + b2] != 0.25f) return false;
    }
    return true;
}

// Stripe table of k_dp_tile for one segment (npr_kernel_tile.hip): the lattice columns 0..lX cut into stripes of 64*R
// columns; per stripe the first / last anti-diagonal on which the band has cells in it and the index of its first row in
// the task's scratch (one row per anti-diagonal of a stripe).  out[0] is the header {stripes, rows}.
void build_stripes(const Segment &s, int R, Stripe *out, int64_t *rows_out) {
    const int64_t S = (s.xe - s.xs) / (64 * R) + 1;
    int64_t rows;
    if (out) {
        rows = stripe_ranges(s.lo.data(), s.n.data(), s.D(), s.xe - s.xs, R, &out[1].df, &out[1].dl, static_cast<int>(sizeof(Stripe) / sizeof(int32_t)));
        stripe_fill(out, s.xe - s.xs, R);
    } else {
        thread_local std::vector<int32_t> df, dl;
        df.resize(S), dl.resize(S);
        rows = stripe_ranges(s.lo.data(), s.n.data(), s.D(), s.xe - s.xs, R, df.data(), dl.data(), 1);
    }
    if (rows_out) *rows_out = rows;
}
inline int64_t stripes_of(const Segment &s, int R) { return (s.xe - s.xs) / (64 * R) + 1; }
}  // namespace

// --------------------------------------------------------------------------------------------------
// batch
// --------------------------------------------------------------------------------------------------

namespace {
int32_t rescore_stage(npr_batch *b);  // NPR_MODE_RESCORE_ORIGINAL: the guide's M columns as a device table (below, with the finish stages)
}

int32_t npr_batch_create(npr_ctx *ctx, const npr_params *params, int64_t n_reads, int64_t n_refs,
                         const uint8_t *ref, const int64_t *ref_off, const int32_t *ref_index,
                         const uint8_t *read, const int64_t *read_off, const int32_t *guide_ops,
                         const int64_t *guide_off, const int32_t *model_slot, npr_batch **out) {
    return npr_batch_create_at(ctx, params, n_reads, n_refs, ref, ref_off, ref_index, read, read_off, guide_ops, guide_off,
                               nullptr, model_slot, out);
}

// Row offsets of the generic kernel (rows padded to 4 cells), made on the device from the band rows the first time a
// generic launch needs them: batches whose tasks all go to the register kernels never pay for them.
static int32_t ensure_coff(npr_batch *b) {
    npr_ctx *ctx = b->ctx;
    if (b->d_coff.p || b->d_lo.count == 0) return NPR_OK;
    const hipError_t e = b->d_coff.alloc_from(ctx, b->d_lo.count);
    if (e != hipSuccess) return fail(ctx, NPR_ERR_NOMEM, "generic row offsets: hipMalloc", e);
    CoffArgs ca{static_cast<int32_t>(b->d_pseg.count), b->d_pseg.p, b->d_n.p, b->d_coff.p};
    const int rc = launch_plan_coff(ca, ctx->stream);
    if (rc != 0) return fail(ctx, NPR_ERR_HIP, "k_plan_coff launch", static_cast<hipError_t>(rc));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return NPR_OK;
}

static int32_t batch_create_at_impl(npr_ctx *ctx, const npr_params *params, int64_t n_reads, int64_t n_refs,
                                    const uint8_t *ref, const int64_t *ref_off, const int32_t *ref_index,
                                    const uint8_t *read, const int64_t *read_begin, const int64_t *read_end, const int32_t *guide_ops,
                                    const int64_t *guide_off, const int64_t *guide_start, const int32_t *model_slot,
                                    npr_batch **out);

// no exception crosses the C ABI: allocation failures of the host stages come back as NPR_ERR_NOMEM
int32_t npr_batch_create_at(npr_ctx *ctx, const npr_params *params, int64_t n_reads, int64_t n_refs,
                            const uint8_t *ref, const int64_t *ref_off, const int32_t *ref_index,
                            const uint8_t *read, const int64_t *read_off, const int32_t *guide_ops,
                            const int64_t *guide_off, const int64_t *guide_start, const int32_t *model_slot,
                            npr_batch **out) {
    try {
        return batch_create_at_impl(ctx, params, n_reads, n_refs, ref, ref_off, ref_index, read, read_off, read_off ? read_off + 1 : nullptr, guide_ops,
                                    guide_off, guide_start, model_slot, out);
    } catch (const std::exception &) {
        return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: out of host memory");
    }
}

int32_t npr_batch_create_spans(npr_ctx *ctx, const npr_params *params, int64_t n_reads, int64_t n_refs,
                               const uint8_t *ref, const int64_t *ref_off, const int32_t *ref_index,
                               const uint8_t *read, const int64_t *read_begin, const int64_t *read_end,
                               const int32_t *guide_ops, const int64_t *guide_off, const int64_t *guide_start,
                               const int32_t *model_slot, npr_batch **out) {
    try {
        return batch_create_at_impl(ctx, params, n_reads, n_refs, ref, ref_off, ref_index, read, read_begin, read_end, guide_ops, guide_off,
                                    guide_start, model_slot, out);
    } catch (const std::exception &) {
        return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: out of host memory");
    }
}

static int32_t batch_create_at_impl(npr_ctx *ctx, const npr_params *params, int64_t n_reads, int64_t n_refs,
                                    const uint8_t *ref, const int64_t *ref_off, const int32_t *ref_index,
                                    const uint8_t *read, const int64_t *read_begin, const int64_t *read_end, const int32_t *guide_ops,
                                    const int64_t *guide_off, const int64_t *guide_start, const int32_t *model_slot,
                                    npr_batch **out) {
    if (!ctx || !params || !out || n_reads < 0 || n_refs < 0) return NPR_ERR_INVALID;
    if (!ref_index && n_refs != n_reads) return fail(ctx, NPR_ERR_INVALID, "npr_batch_create: without ref_index, n_refs must equal n_reads");
    auto ref_of = [&](int64_t i) -> int64_t { return ref_index ? ref_index[i] : i; };
    if (n_reads > 0 && (!ref_off || !read_begin || !read_end || !guide_off)) return fail(ctx, NPR_ERR_INVALID, "npr_batch_create: null offsets");
    *out = nullptr;
    std::unique_ptr<npr_batch> b(new (std::nothrow) npr_batch);
    if (!b) return NPR_ERR_NOMEM;
    b->ctx = ctx;
    // Every error return below may leave copies and planner kernels queued on the context's streams that read or write
    // buffers of this batch (and the context's pinned staging): released buffers go to the context's cache, not to hipFree
    // (which would synchronise), so nothing may still be in flight when they do.  Declared after `b`: runs before its
    // destructor.
    struct DrainOnError {
        npr_ctx *c;
        bool armed = true;
        ~DrainOnError() {
            if (!armed) return;
            (void)hipStreamSynchronize(c->side[0]);
            (void)hipStreamSynchronize(c->stream);
        }
    } drain{ctx};
    b->params = *params;
    if (b->params.max_pairs_per_base <= 0) b->params.max_pairs_per_base = 6;
    b->n_reads = n_reads;
    b->ref_len.resize(n_reads);
    b->read_len.resize(n_reads);
    b->read_status.assign(n_reads, NPR_OK);
    b->gstart.assign(2 * n_reads, 0);
    b->ref_id.resize(n_reads);
    for (int64_t i = 0; i < n_reads; ++i) b->ref_id[i] = static_cast<int32_t>(ref_of(i));
    b->read_first_task.assign(n_reads, 0);
    b->read_ntasks.assign(n_reads, 0);
    b->guide_off.assign(guide_off, guide_off + (n_reads ? n_reads + 1 : 0));
    // the guides themselves are needed again only where the result IS the guide (--rescoreOriginalAlignment); copying
    // them for every realign batch cost 35 ms of a north-star batch's 80 (240 MB, one thread, first touch)
    if (n_reads && b->params.mode == NPR_MODE_RESCORE_ORIGINAL) b->guide_ops.assign(guide_ops, guide_ops + 2 * guide_off[n_reads]);

    StageTimer tm("batch_create");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e;
    // 1. Host, O(cigar operations) per read: the guide's window, validation, matrix splits and the plan points of every
    // segment (npr_host.cpp plan_points).  Worker threads take chunks of reads and append to their chunk's plan.
    constexpr int64_t kChunk = 32;
    const int64_t nchunks = (n_reads + kChunk - 1) / kChunk;
    std::vector<PointPlan> chunk_plan(nchunks);
    parallel_for(nchunks, ctx->host_threads, [&](int64_t c) {
        PointPlan &pp = chunk_plan[c];
        for (int64_t i = c * kChunk, hi = std::min(n_reads, (c + 1) * kChunk); i < hi; ++i) {
            const int64_t k = ref_of(i);
            if (k < 0 || k >= n_refs) {
                b->ref_len[i] = b->read_len[i] = 0;
                b->read_status[i] = NPR_ERR_INVALID;
                continue;
            }
            int64_t lX = ref_off[k + 1] - ref_off[k], lY = read_end[i] - read_begin[i];
            int32_t rc = lY < 0 ? NPR_ERR_INVALID : NPR_OK;
            if (guide_start) {  // the window the guide covers
                const int64_t gx = guide_start[2 * i], gy = guide_start[2 * i + 1];
                int64_t sx = 0, sy = 0;
                for (int64_t q = guide_off[i]; q < guide_off[i + 1]; ++q) {
                    const int32_t op = guide_ops[2 * q], len = guide_ops[2 * q + 1];
                    if (len < 0) rc = NPR_ERR_INVALID;
                    if (op == NPR_OP_M || op == NPR_OP_D) sx += len;
                    if (op == NPR_OP_M || op == NPR_OP_I) sy += len;
                }
                if (gx < 0 || gy < 0 || gx + sx > lX || gy + sy > lY) rc = NPR_ERR_INVALID;
                b->gstart[2 * i] = gx, b->gstart[2 * i + 1] = gy;
                lX = sx, lY = sy;
            }
            b->ref_len[i] = lX;
            b->read_len[i] = lY;
            const int32_t slot = model_slot ? model_slot[i] : 0;
            if (slot < 0 || slot >= NPR_MAX_MODELS || !ctx->model_set[slot]) rc = NPR_ERR_MODEL;
            const size_t seg0 = pp.segs.size(), pt0 = pp.points.size();
            if (rc == NPR_OK) rc = plan_points(b->params, lX, lY, guide_ops + 2 * guide_off[i], guide_off[i + 1] - guide_off[i], pp);
            if (rc != NPR_OK) {
                pp.segs.resize(seg0), pp.points.resize(pt0);
                b->ref_len[i] = b->read_len[i] = 0;
            }
            for (size_t q = seg0; q < pp.segs.size(); ++q) pp.segs[q].owner = i;
            b->read_ntasks[i] = static_cast<int32_t>(pp.segs.size() - seg0);
            b->read_status[i] = rc;
        }
    });
    tm.lap("plan points");

    // 2. flatten: segments in read order, their points and band rows at prefix offsets
    std::vector<int64_t> chunk_seg0(nchunks + 1, 0), chunk_pt0(nchunks + 1, 0);
    for (int64_t c = 0; c < nchunks; ++c) {
        chunk_seg0[c + 1] = chunk_seg0[c] + static_cast<int64_t>(chunk_plan[c].segs.size());
        chunk_pt0[c + 1] = chunk_pt0[c] + static_cast<int64_t>(chunk_plan[c].points.size());
    }
    const int64_t ntasks = chunk_seg0[nchunks], npoints = chunk_pt0[nchunks];
    if (ntasks >= (int64_t(1) << 31)) return fail(ctx, NPR_ERR_INVALID, "npr_batch_create: too many tasks");
    {
        int64_t first = 0;
        for (int64_t i = 0; i < n_reads; ++i) b->read_first_task[i] = static_cast<int32_t>(first), first += b->read_ntasks[i];
    }
    // the read's windows as they stand in the caller's buffers (ASCII), reference part then read part, encoded on the device
    std::vector<int64_t> win_off(n_reads + 1, 0);
    for (int64_t i = 0; i < n_reads; ++i) win_off[i + 1] = win_off[i] + (b->read_ntasks[i] ? b->ref_len[i] + b->read_len[i] : 0);
    const int64_t seq_bytes = win_off[n_reads];
    std::vector<SegPlan> seg(ntasks);  // flat, read order
    std::vector<PlanSeg> pseg(ntasks);
    int64_t band_entries = 0;
    for (int64_t c = 0; c < nchunks; ++c)
        for (size_t q = 0; q < chunk_plan[c].segs.size(); ++q) {
            const int64_t k = chunk_seg0[c] + static_cast<int64_t>(q);
            seg[k] = chunk_plan[c].segs[q];
            PlanSeg &ps = pseg[k];
            ps.point_first = chunk_pt0[c] + seg[k].point_first;
            ps.band_off = band_entries;
            ps.pieces = seg[k].pieces;
            ps.lX = static_cast<int32_t>(seg[k].xe - seg[k].xs), ps.lY = static_cast<int32_t>(seg[k].ye - seg[k].ys), ps.pad = 0;
            band_entries += static_cast<int64_t>(ps.lX) + ps.lY + 1;
        }
    // pinned staging (kept by the context): plan points, then the sequence windows
    const size_t stage_pts = (static_cast<size_t>(npoints) * sizeof(PlanPoint) + 255) & ~size_t(255);
    const size_t stage_need = stage_pts + static_cast<size_t>(seq_bytes) + 256;
    if (stage_need > ctx->pin_stage_bytes) {
        if (ctx->pin_stage) (void)hipHostFree(ctx->pin_stage);
        ctx->pin_stage = nullptr, ctx->pin_stage_bytes = 0;
        if ((e = hipHostMalloc(&ctx->pin_stage, stage_need + stage_need / 4, hipHostMallocDefault)) != hipSuccess)
            return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: hipHostMalloc", e);
        ctx->pin_stage_bytes = stage_need + stage_need / 4;
    }
    PlanPoint *const h_points = static_cast<PlanPoint *>(ctx->pin_stage);
    uint8_t *const h_seq = static_cast<uint8_t *>(ctx->pin_stage) + stage_pts;
    parallel_for(nchunks, ctx->host_threads, [&](int64_t c) {
        if (!chunk_plan[c].points.empty())
            std::memcpy(h_points + chunk_pt0[c], chunk_plan[c].points.data(), chunk_plan[c].points.size() * sizeof(PlanPoint));
        for (int64_t i = c * kChunk, hi = std::min(n_reads, (c + 1) * kChunk); i < hi; ++i) {
            if (!b->read_ntasks[i]) continue;
            std::memcpy(h_seq + win_off[i], ref + ref_off[ref_of(i)] + b->gstart[2 * i], static_cast<size_t>(b->ref_len[i]));
            std::memcpy(h_seq + win_off[i] + b->ref_len[i], read + read_begin[i] + b->gstart[2 * i + 1], static_cast<size_t>(b->read_len[i]));
        }
    });
    chunk_plan.clear();
    tm.lap("flatten + stage");

    // 3. device: band rows of every anti-diagonal, per-segment summaries
    DevBuf<PlanPoint> d_points;
    DevBuf<SegSummary> d_summary;
    if ((e = d_points.alloc_from(ctx, npoints)) != hipSuccess || (e = b->d_pseg.alloc_from(ctx, ntasks)) != hipSuccess || (e = d_summary.alloc_from(ctx, ntasks)) != hipSuccess ||
        (e = b->d_lo.alloc_from(ctx, band_entries + 16)) != hipSuccess || (e = b->d_n.alloc_from(ctx, band_entries + 16)) != hipSuccess ||  // (+16: the schedule's walkers read rows four at a time, up to eight past a segment's last)
        (e = b->d_seq.alloc_from(ctx, seq_bytes + 16)) != hipSuccess)
        return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: hipMalloc", e);
    std::vector<SegSummary> summary(ntasks);
    if (ntasks) {
        HIP_TRY(ctx, hipMemcpyAsync(d_points.p, h_points, d_points.bytes(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(b->d_pseg.p, pseg.data(), b->d_pseg.bytes(), hipMemcpyHostToDevice, ctx->stream));
        PlanArgs pa{static_cast<int32_t>(ntasks), b->params.band_mode == NPR_BAND_FIXED ? 1 : 0,
                    b->params.band_mode == NPR_BAND_FIXED ? b->params.fixed_width / 2 : b->params.diagonal_expansion,
                    d_points.p, b->d_pseg.p, b->d_lo.p, b->d_n.p, d_summary.p};
        int rc = launch_plan_bands(pa, ctx->stream);
        if (rc != 0) return fail(ctx, NPR_ERR_HIP, "k_plan_bands launch", static_cast<hipError_t>(rc));
        HIP_TRY(ctx, hipMemcpyAsync(summary.data(), d_summary.p, d_summary.bytes(), hipMemcpyDeviceToHost, ctx->stream));
        // the sequences travel and are encoded while the host looks at the summaries
        if (seq_bytes) {
            HIP_TRY(ctx, hipMemcpyAsync(b->d_seq.p, h_seq, static_cast<size_t>(seq_bytes), hipMemcpyHostToDevice, ctx->side[0]));
            if ((rc = launch_encode(b->d_seq.p, seq_bytes, ctx->side[0])) != 0) return fail(ctx, NPR_ERR_HIP, "k_encode launch", static_cast<hipError_t>(rc));
            HIP_TRY(ctx, hipEventRecord(ctx->side_done[0], ctx->side[0]));
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    tm.lap("device band rows");
    for (int64_t k = 0; k < ntasks; ++k)
        if (summary[k].max_width > (1 << 22) || summary[k].cells >= (int64_t(1) << 40)) b->read_status[seg[k].owner] = NPR_ERR_BAND_TOO_WIDE;
    // (a read refused here keeps its tasks -- they are cheap to run and its status says the results do not count)

    // 4. kernel classes.  The register kernels on a frame that follows the anti-diagonal take bands whose frame schedule
    // exists, tried from the smallest frame up (on the device: the schedule is sequential per segment); bands too wide for
    // one wavefront's frame go to the stripe kernel (k_dp_tile), whatever their shape.  A batch staged for the E-step
    // (NPR_MODE_EXPECTATIONS) keeps the classes that have an E-step kernel.
    const bool force_generic = ctx->opt[NPR_OPT_KERNEL] == 1;  // no register kernel (A/B runs, tests)
    const int lds_max_w = generic_max_wcap();
    const bool no_wide = ctx->opt[NPR_OPT_NO_WIDE] != 0;  // no multi-wavefront register kernel (A/B runs, tests)
    const int cmin = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(kSchedClasses, ctx->opt[NPR_OPT_CLASS_MIN])));  // bring-up: smallest register class to use
    const bool use_tile = !force_generic && ctx->opt[NPR_OPT_NO_TILE] == 0;  // (E-step batches too: k_em_tile)
    std::vector<uint32_t> cand(ntasks, 0);
    std::vector<int64_t> sched_off(ntasks, -1);
    // (the first task's words start kCtlFrontPad rows into d_ctl: the backward sweep of k_dp_rs reads its control words up to
    // three rows below the one it is on, row 0 included, without a clamp)
    constexpr int64_t kCtlFrontPad = 4;
    int64_t ctl_entries = kCtlFrontPad;
    for (int64_t k = 0; k < ntasks; ++k) {
        if (force_generic) break;
        for (int c = cmin; c < kSchedClasses; ++c) {
            if (kClassTab[c].kind == K_WIDE && (use_tile || no_wide)) continue;
            if (kClassTab[c].kind == K_STAIR && !stair_fits(static_cast<int64_t>(pseg[k].lX) + pseg[k].lY + 1, kClassTab[c].slots())) continue;
            if (summary[k].max_width <= stair_max_width(kClassTab[c].R, kClassTab[c].NW)) cand[k] |= 1u << c;
        }
        if (cand[k]) sched_off[k] = ctl_entries, ctl_entries += static_cast<int64_t>(pseg[k].lX) + pseg[k].lY + 1;
    }
    std::vector<int32_t> sched_cls(ntasks, -1);
    std::vector<int64_t> sched_cells(ntasks, 0);
    if ((e = b->d_ctl.alloc_from(ctx, 2 * ctl_entries + 16)) != hipSuccess)  // (+16: k_dp_rs reads its control words two rows ahead, k_dp_mid_rs up to six)
        return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: hipMalloc", e);
    if (ctl_entries > kCtlFrontPad) {
        DevBuf<uint32_t> d_cand;
        DevBuf<int64_t> d_off, d_cells;
        DevBuf<int32_t> d_cls;
        if ((e = d_cand.alloc_from(ctx, ntasks)) != hipSuccess || (e = d_off.alloc_from(ctx, ntasks)) != hipSuccess || (e = d_cells.alloc_from(ctx, ntasks)) != hipSuccess ||
            (e = d_cls.alloc_from(ctx, ntasks)) != hipSuccess)
            return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: hipMalloc", e);
        HIP_TRY(ctx, hipMemcpyAsync(d_cand.p, cand.data(), d_cand.bytes(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_off.p, sched_off.data(), d_off.bytes(), hipMemcpyHostToDevice, ctx->stream));
        SchedArgs sa{static_cast<int32_t>(ntasks), b->d_pseg.p, d_summary.p, b->d_lo.p, b->d_n.p, d_off.p, d_cand.p, b->d_ctl.p, d_cls.p, d_cells.p};
        // the walk of a segment in chunks that compose (npr_plan.hip): chunk tables
        std::vector<int64_t> chunk_off(ntasks + 1, 0);
        uint32_t cand_union = 0;
        for (int64_t k = 0; k < ntasks; ++k) {
            chunk_off[k + 1] = chunk_off[k] + (cand[k] ? plan_sched_chunks_of(static_cast<int64_t>(pseg[k].lX) + pseg[k].lY) : 0);
            cand_union |= cand[k];
        }
        const int64_t n_chunks = chunk_off[ntasks];
        DevBuf<int64_t> d_chunk_off;
        DevBuf<uint8_t> d_chunks;
        DevBuf<int32_t> d_cur;
        if ((e = d_chunk_off.alloc_from(ctx, ntasks + 1)) != hipSuccess || (e = d_chunks.alloc_from(ctx, plan_sched_chunk_bytes(n_chunks))) != hipSuccess ||
            (e = d_cur.alloc_from(ctx, ntasks + kSchedClasses)) != hipSuccess)
            return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: hipMalloc", e);
        HIP_TRY(ctx, hipMemcpyAsync(d_chunk_off.p, chunk_off.data(), d_chunk_off.bytes(), hipMemcpyHostToDevice, ctx->stream));
        const int rc = launch_plan_sched(sa, d_chunk_off.p, n_chunks, d_chunks.p, d_cur.p, cand_union, ctx->stream);
        if (rc != 0) return fail(ctx, NPR_ERR_HIP, "k_plan_sched launch", static_cast<hipError_t>(rc));
        HIP_TRY(ctx, hipMemcpyAsync(sched_cls.data(), d_cls.p, d_cls.bytes(), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(sched_cells.data(), d_cells.p, d_cells.bytes(), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    tm.lap("device frame schedules");
    std::vector<int8_t> cls_of(ntasks);
    std::vector<int32_t> tile_list;
    std::vector<int64_t> tile_off_of(ntasks, -1), tile_offs;
    int64_t stripe_entries = 0;
    bool any_generic = false;
    for (int64_t k = 0; k < ntasks; ++k) {
        int c = sched_cls[k];
        if (c < 0) {
            const int64_t w = summary[k].max_width;
            c = use_tile ? kTileClass : (w <= 512 ? kFirstGeneric : (w <= 1024 ? kFirstGeneric + 1 : (w <= lds_max_w ? kFirstGeneric + 2 : kFirstGeneric + 3)));
        }
        cls_of[k] = static_cast<int8_t>(c);
        // the stripe kernels address a stripe's rows (1 KiB each) with a 32-bit byte offset behind one descriptor: a stripe of
        // 2^21 rows or more would wrap.  No stripe has more rows than its task has anti-diagonals.
        if (kClassTab[c].kind == K_TILE && static_cast<int64_t>(pseg[k].lX) + pseg[k].lY + 1 >= (int64_t(1) << 21))
            b->read_status[seg[k].owner] = NPR_ERR_BAND_TOO_WIDE;
        if (kClassTab[c].kind == K_TILE) {
            tile_list.push_back(static_cast<int32_t>(k));
            tile_off_of[k] = stripe_entries;
            tile_offs.push_back(stripe_entries);
            stripe_entries += 1 + pseg[k].lX / (64 * kClassTab[c].R) + 1;
        }
        any_generic |= kClassTab[c].kind == K_GENERIC_LDS || kClassTab[c].kind == K_GENERIC_GLOBAL;
    }
    // The one-wavefront frame tasks run in row-scaled arithmetic (npr_rs.h) -- every one of them, provided the loaded models let a row's
    // values be renormalised every NPR_RS_K anti-diagonals (rs_model_ok); a task for which one exponent per row turns out not to be
    // enough says so and npr_batch_run runs it again in class 0-2's kernel.  NPR_OPT_ARITH = 1: none (the per-cell-exponent kernels
    // throughout, A/B).
    {
        bool rs = ctx->opt[NPR_OPT_ARITH] != 1 && !force_generic && b->params.mode != NPR_MODE_EXPECTATIONS;
        for (int sl = 0; sl < NPR_MAX_MODELS; ++sl)
            if (ctx->model_set[sl] && !rs_model_ok(ctx->models[sl])) rs = false;
        b->pair_rs = rs;
        if (rs)
            for (int64_t k = 0; k < ntasks; ++k) {
                if (cls_of[k] >= 0 && cls_of[k] < 3) cls_of[k] = static_cast<int8_t>(kFirstRs + cls_of[k]);
                // the stripe tasks run in column-scaled arithmetic (k_dp_tile_cs, round 6: one exponent per lane of a stripe; same bits, and a
                // per-lane range certificate that the reference's 3000-cell-wide rectangles pass -- DESIGN.md 5.1f); NPR_OPT_TILE_RS = 2: the
                // per-cell-exponent k_dp_tile throughout (A/B)
                else if (cls_of[k] == kTileClass && ctx->opt[NPR_OPT_TILE_RS] != 2) cls_of[k] = static_cast<int8_t>(kTileRsClass);
            }
    }
    // A read on ONE wavefront is a serial chain of 2 * (lX + lY) steps: a launch lasts at least as long as its longest task, and a class
    // with fewer tasks than the chip has wavefront slots leaves the rest idle.  k_dp_mid_rs (classes 12-14, round 5) runs a task's two
    // sweeps on two wavefronts that meet in the middle: half the chain for the bytes and instructions of k_dp_rs, so EVERY row-scaled
    // task of MID_MIN_D anti-diagonals or more goes there (a 1/8 shard of configs[3]: DP launch 41.7 -> 28.5 ms, configs[1] 1.27 -> 0.75 ms,
    // the headline batch 138.9 -> 131.6 ms with round 5's other changes); shorter ones stay with k_dp_rs.  (Rounds 3-4 had kernels with both
    // sweeps whole and a third pass over the rows of both, k_dp_pair / k_dp_pair_rs, for classes that filled at most half of the chip.)
    // NPR_OPT_PAIR 1: never; 2: only the tasks longer than a wavefront's fair share of their class, as far as second wavefronts are free;
    // 0 / 3: every task.
    bool any_pair = false;
    {
        const int64_t pe = ctx->opt[NPR_OPT_PAIR];
        const bool pair_off = pe == 1, pair_long = pe == 2;
        if (b->pair_rs && !pair_off)
            for (int c = 0; c < 3; ++c) {
                std::vector<int32_t> mine;
                int64_t cost = 0;
                for (int64_t k = 0; k < ntasks; ++k)
                    if (cls_of[k] == kFirstRs + c) mine.push_back(static_cast<int32_t>(k)), cost += static_cast<int64_t>(pseg[k].lX) + pseg[k].lY + 1;
                if (mine.empty()) continue;
                const int64_t slots = static_cast<int64_t>(ctx->cu_count) * mid_waves_per_cu(kClassTab[c].R);
                const int64_t n = static_cast<int64_t>(mine.size()), fair = cost / slots;
                int64_t room = !pair_long ? n : (n < slots ? slots - n : n);  // second wavefronts to be had
                std::sort(mine.begin(), mine.end(), [&](int32_t x, int32_t y) { return pseg[x].lX + pseg[x].lY > pseg[y].lX + pseg[y].lY; });
                for (int32_t k : mine) {
                    const int64_t len = static_cast<int64_t>(pseg[k].lX) + pseg[k].lY + 1;
                    if (room <= 0 || (pair_long && (len <= fair || len < 256))) break;
                    if (len - 1 < MID_MIN_D) break;  // (sorted by length: the rest is shorter still; k_dp_mid_rs needs a block on either side of its cut)
                    cls_of[k] = static_cast<int8_t>(kFirstPair + c), --room, any_pair = true;
                }
            }
    }
    // k_dp_tile tasks are ordered by the forward scratch they need (one row per anti-diagonal of a stripe: also what a
    // task costs): a workgroup's region is sized by its FIRST task, every later one from the queue is smaller
    std::vector<int64_t> tile_need(ntasks, 0), rowmask_off_of(ntasks, -1);
    if ((e = b->d_stripes.alloc_from(ctx, stripe_entries)) != hipSuccess) return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: hipMalloc", e);
    if (!tile_list.empty()) {
        DevBuf<int32_t> d_list;
        DevBuf<int64_t> d_toff, d_rows;
        const size_t nt = tile_list.size();
        if ((e = d_list.alloc_from(ctx, nt)) != hipSuccess || (e = d_toff.alloc_from(ctx, nt)) != hipSuccess || (e = d_rows.alloc_from(ctx, nt)) != hipSuccess)
            return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: hipMalloc", e);
        HIP_TRY(ctx, hipMemcpyAsync(d_list.p, tile_list.data(), d_list.bytes(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_toff.p, tile_offs.data(), d_toff.bytes(), hipMemcpyHostToDevice, ctx->stream));
        StripeArgs ta{static_cast<int32_t>(nt), kClassTab[kTileClass].R, d_list.p, b->d_pseg.p, d_summary.p, b->d_lo.p, b->d_n.p, d_toff.p, b->d_stripes.p, d_rows.p};
        const int rc = launch_plan_stripes(ta, ctx->stream);
        if (rc != 0) return fail(ctx, NPR_ERR_HIP, "k_plan_stripes launch", static_cast<hipError_t>(rc));
        std::vector<int64_t> rows(nt);
        HIP_TRY(ctx, hipMemcpyAsync(rows.data(), d_rows.p, d_rows.bytes(), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t q = 0; q < nt; ++q) tile_need[tile_list[q]] = (tile_scratch_cells(rows[q], kClassTab[kTileClass].R) + 63) & ~int64_t(63);
        // the lane masks of all those rows, one word each
        std::vector<int64_t> moff(nt);
        int64_t mask_rows = 0;
        for (size_t q = 0; q < nt; ++q) moff[q] = mask_rows, rowmask_off_of[tile_list[q]] = mask_rows, mask_rows += rows[q];
        DevBuf<int64_t> d_moff;
        if ((e = d_moff.alloc_from(ctx, nt)) != hipSuccess || (e = b->d_rowmask.alloc_from(ctx, mask_rows)) != hipSuccess)
            return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: hipMalloc", e);
        HIP_TRY(ctx, hipMemcpyAsync(d_moff.p, moff.data(), d_moff.bytes(), hipMemcpyHostToDevice, ctx->stream));
        RowMaskArgs ma{static_cast<int32_t>(nt), d_list.p, b->d_pseg.p, b->d_lo.p, b->d_n.p, d_toff.p, b->d_stripes.p, d_moff.p, b->d_rowmask.p};
        const int rc2 = launch_plan_rowmask(ma, ctx->stream);
        if (rc2 != 0) return fail(ctx, NPR_ERR_HIP, "k_plan_rowmask launch", static_cast<hipError_t>(rc2));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // d_list / d_toff / d_moff go out of scope
    }
    if (any_generic) {
        const int32_t rc = ensure_coff(b.get());
        if (rc != NPR_OK) return rc;
    }
    tm.lap("device stripe tables");

    // 5. tasks, grouped by class, the costliest first
    // (the frame kernels' tasks by the forward scratch they need, which is what they cost too: a workgroup's scratch region
    // may then be sized by its FIRST task, as the stripe kernel's are -- everything the queue hands it later is smaller)
    std::vector<int64_t> pad_of(ntasks);
    for (int64_t k = 0; k < ntasks; ++k) pad_of[k] = std::max(summary[k].generic_cells, is_register_class(cls_of[k]) ? sched_cells[k] : 0);  // either kernel may run the task
    std::vector<int32_t> rank(ntasks);
    std::iota(rank.begin(), rank.end(), 0);
    std::stable_sort(rank.begin(), rank.end(), [&](int32_t a, int32_t c) {
        if (cls_of[a] != cls_of[c]) return cls_of[a] < cls_of[c];
        if (tile_need[a] != tile_need[c]) return tile_need[a] > tile_need[c];
        if (is_register_class(cls_of[a]) && pad_of[a] != pad_of[c]) return pad_of[a] > pad_of[c];
        return summary[a].cells > summary[c].cells;
    });
    b->task_of.assign(ntasks, 0);
    for (int64_t k = 0; k < ntasks; ++k) b->task_of[rank[k]] = static_cast<int32_t>(k);
    b->tasks.resize(ntasks);
    b->task_cells.resize(ntasks);
    int64_t pair_total = 0, max_pad = 0, max_width = 0, total_cells = 0;
    int64_t cls_count[kClasses] = {}, cls_width[kClasses] = {}, cls_cells[kClasses] = {};
    for (int64_t k = 0; k < ntasks; ++k) {
        const int32_t g = rank[k];
        const SegPlan &s = seg[g];
        const int64_t i = s.owner;
        Task &t = b->tasks[k];
        t.x_off = win_off[i] + s.xs;
        t.y_off = win_off[i] + b->ref_len[i] + s.ys;
        t.band_off = pseg[g].band_off;
        t.lX = pseg[g].lX;
        t.lY = pseg[g].lY;
        t.D = t.lX + t.lY;
        t.flags = (s.ragged_start ? 1 : 0) | (s.ragged_end ? 2 : 0);
        t.model = model_slot ? model_slot[i] : 0;
        t.xs = static_cast<int32_t>(s.xs);
        t.ys = static_cast<int32_t>(s.ys);
        t.read = static_cast<int32_t>(i);
        const int64_t cells = summary[g].cells;
        const int64_t cap = std::min<int64_t>(cells, static_cast<int64_t>(b->params.max_pairs_per_base) * std::min(t.lX, t.lY) + 64);
        t.pair_cap = static_cast<int32_t>(std::min<int64_t>(cap, INT32_MAX));
        t.pair_off = pair_total;
        pair_total += t.pair_cap;
        b->task_cells[k] = cells;
        total_cells += cells;
        max_width = std::max<int64_t>(max_width, summary[g].max_width);
        const int c = cls_of[g];
        t.ctl_off = is_register_class(c) ? sched_off[g] : -1;
        t.tile_off = tile_off_of[g];
        t.rowmask_off = rowmask_off_of[g];
        const int64_t pad = pad_of[g];
        if (pad >= (int64_t(1) << 32)) return fail(ctx, NPR_ERR_INVALID, "npr_batch_create: segment too large");
        t.cells_pad = static_cast<int32_t>(std::min<int64_t>(pad, INT32_MAX));
        max_pad = std::max(max_pad, pad);
        ++cls_count[c];
        cls_width[c] = std::max<int64_t>(cls_width[c], summary[g].max_width);
        cls_cells[c] += cells;
    }
    if (seq_bytes) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_done[0], 0));
    tm.lap("tasks");
    // 6. launch geometry and the remaining device buffers
    b->slot_stride = (max_pad + 63) & ~int64_t(63);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
    // (sequences, band rows, control words and stripe tables are allocated already)
    const int64_t fixed = pair_total * 12 + ntasks * (int64_t)(sizeof(Task) + sizeof(TaskOut)) + (any_generic ? 0 : band_entries * 4);
    const size_t arena_now = ctx->arena->cells.load();
    const int64_t budget = static_cast<int64_t>((free_b + ctx->cache_bytes + arena_now * 8) * 0.9) - fixed;
    int64_t fit = INT32_MAX;
    if (b->slot_stride > 0) {
        fit = budget / (b->slot_stride * 8);
        if (fit < 1) return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: not enough device memory for one forward scratch region");
    }
    int64_t max_grid = 1, ring_floats = 0, first = 0;
    for (int c = 0; c < kClasses; ++c) {
        if (!cls_count[c]) continue;
        npr_batch::Launch L{};
        L.cls = c;
        L.first = static_cast<int>(first);
        L.count = static_cast<int>(cls_count[c]);
        L.cells = cls_cells[c];
        L.width = cls_width[c];
        first += cls_count[c];
        int waves_per_cu;
        if (kClassTab[c].kind == K_MID) {  // workgroups of two wavefronts
            waves_per_cu = mid_waves_per_cu(kClassTab[c].R) / 2;
            // NPR_OPT_OVERLAP = 1: half of every SIMD's wavefront slots, and 224 of its 512 registers, left to the staging and MEA kernels of
            // the batches this one runs next to.  A persistent DP launch that fills the chip (7 x 72 registers) leaves room for nothing: every
            // other kernel of the job then waits for the launch's last wavefronts (profiles/r05_c3_job_trace.txt).  Measured on the files ->
            // file job of 50 000 reads, wavefronts per SIMD 7 / 6 / 5 / 4 / 3: 372 / 372 / 371 / 352-361 / 388 ms.
            if (ctx->overlap == 1 && kClassTab[c].R <= 2) waves_per_cu = std::min(waves_per_cu, 8);
            L.wcap = 0;
            L.lds = stair_lds_bytes();
            L.threads = 128;
        } else if (is_one_wave_kind(kClassTab[c].kind)) {  // VGPR-limited: 71 / 80 (held there by amdgpu_waves_per_eu) / 162 registers: 7 / 6 / 3 waves per SIMD
            waves_per_cu = kClassTab[c].kind == K_RS ? rs_waves_per_cu(kClassTab[c].R) : stair_waves_per_cu(kClassTab[c].R);
            if (ctx->overlap == 1 && kClassTab[c].R <= 2) waves_per_cu = std::min(waves_per_cu, 16);  // (four per SIMD, as for the two-wavefront classes above)
            L.wcap = 0;
            L.lds = stair_lds_bytes();
            L.threads = 64;
        } else if (kClassTab[c].kind == K_WIDE) {  // workgroups per CU by VGPRs: 111 (R = 2) -> 4 waves per SIMD, 168-176 (R = 4) -> 2-3
            const int nw = kClassTab[c].NW;
            // workgroups per CU: 111 VGPRs (R = 2) and 128 (4 x 8, held there by amdgpu_waves_per_eu) -> 4 waves per SIMD;
            // 4 x 12: 168 VGPRs, 3 waves per SIMD
            waves_per_cu = (kClassTab[c].R == 2 || nw <= 8) ? 16 / nw : 1;
            L.wcap = 0;
            L.lds = wide_lds_bytes(nw);
            L.threads = 64 * nw;
        } else if (is_tile_kind(kClassTab[c].kind)) {
            // 80 VGPRs: 6 wavefronts per SIMD, 24 per CU, shared by workgroups of NW wavefronts.  A read's band offers a
            // parallelism of about four stripes on average (rectangles of ~1000 columns, each stripe starting 128 + 16..31
            // anti-diagonals after its left neighbour): measured on 8192 x 8 kb reads in the reference's band, 2 / 3 / 4 / 6 / 8
            // wavefronts per task give 1.26 / 1.71 / 2.06 / 1.42 / 1.64e11 cells/s (more tasks in flight need more scratch)
            // (k_dp_tile_cs, round 6, same batch: 2 / 3 / 4 / 6 / 8 wavefronts per task 338 / 281 / 294 / 396 / 365 ms -- its steps are shorter, the
            // hand-overs are not, so a fourth wavefront waits more than it works)
            int nw = kClassTab[c].kind == K_TILE_RS ? 3 : 4;
            if (ctx->opt[NPR_OPT_TILE_WAVES] > 0) nw = static_cast<int>(std::min<int64_t>(8, ctx->opt[NPR_OPT_TILE_WAVES]));
            waves_per_cu = std::max(1, 24 / nw);
            L.wcap = nw;
            L.lds = kClassTab[c].kind == K_TILE_RS ? tile_cs_lds_bytes(nw) : tile_lds_bytes(nw);
            L.threads = 64 * nw;
        } else if (kClassTab[c].kind == K_GENERIC_LDS) {
            // several wavefronts per task: these tasks are big, their forward scratch caps how many can be
            // resident, and one wavefront each would leave the SIMDs idle
            L.wcap = static_cast<int>((std::max<int64_t>(cls_width[c], 64) + 3) & ~int64_t(3));
            L.lds = generic_lds_bytes(L.wcap);
            const int wg_per_cu = std::max<int>(1, static_cast<int>((160 * 1024) / (L.lds + 256)));
            L.threads = wg_per_cu >= 2 ? 256 : 512;                     // a lone workgroup on a CU gets 8 wavefronts
            waves_per_cu = std::min(wg_per_cu, 2048 / L.threads);        // workgroups per CU
        } else {
            L.wcap = static_cast<int>((cls_width[c] + 3) & ~int64_t(3));
            L.lds = generic_lds_bytes(0);
            L.threads = 512;
            waves_per_cu = 2;  // workgroups per CU
        }
        if (ctx->opt[NPR_OPT_WAVES_PER_CU] > 0) waves_per_cu = static_cast<int>(std::min<int64_t>(64, ctx->opt[NPR_OPT_WAVES_PER_CU]));
        int64_t grid = std::min<int64_t>(L.count, static_cast<int64_t>(ctx->cu_count) * waves_per_cu);
        L.grid = static_cast<int>(std::max<int64_t>(1, grid));
        if (std::getenv("NPR_TIMING"))
            std::fprintf(stderr, "[npr] class %d (kind %d R %d NW %d): %lld tasks, %lld cells, widest %lld, grid %d x %d threads\n", c,
                         kClassTab[c].kind, kClassTab[c].R, kClassTab[c].NW, (long long)cls_count[c], (long long)cls_cells[c],
                         (long long)cls_width[c], L.grid, L.threads);
        b->launches.push_back(L);
    }
    // The launches run concurrently, each on its own scratch regions: the regions of all of them must fit.  Uniform regions
    // of slot_stride cells (the largest task of the batch) for the generic / multi-wavefront launches, and for the
    // one-wavefront frame launches of a small batch; the stripe launch one region per workgroup, sized by the workgroup's
    // first task (its tasks are sorted by need, so everything the queue hands out later is smaller) -- and so the
    // one-wavefront frame launches of a big realign batch (round 3): 6144 uniform regions sized for the one 20 kb read of a
    // config-3 chunk took 252 GB where the reads that actually start in them need 130, which is what lets a pipelined job keep
    // three batches on the device.  (Not for batches staged for the E-step, whose kernels index the planes of a region by
    // slot_stride; npr_batch_expectations refuses a batch laid out this way.)
    npr_batch::Launch *tileL = nullptr;
    for (auto &L : b->launches)
        if (is_tile_kind(kClassTab[L.cls].kind)) tileL = &L;
    const int64_t tile_min = tileL ? tile_need[rank[tileL->first]] : 0;
    int64_t stair_grid = 0;
    for (auto &L : b->launches)
        if (is_one_wave_kind(kClassTab[L.cls].kind)) stair_grid += L.grid;
    int64_t var_min_bytes = int64_t(32) << 30;  // uniform stair scratch above this goes variable (NPR_OPT_VARIABLE_SCRATCH: 1 always, 2 never; tests)
    if (ctx->opt[NPR_OPT_VARIABLE_SCRATCH] == 1) var_min_bytes = 0;
    if (ctx->opt[NPR_OPT_VARIABLE_SCRATCH] == 2) var_min_bytes = int64_t(1) << 60;
    b->variable_regions = b->params.mode != NPR_MODE_EXPECTATIONS && stair_grid > 0 && stair_grid * b->slot_stride * 8 >= var_min_bytes &&
                          !force_generic;
    if (any_pair) b->variable_regions = true;  // (their regions hold two sets of rows: not a layout the E-step kernels know)
    auto uniform = [&](const npr_batch::Launch &L) {
        return &L != tileL && kClassTab[L.cls].kind != K_MID && !(b->variable_regions && is_one_wave_kind(kClassTab[L.cls].kind));
    };
    int64_t sum_grid = 0;
    for (auto &L : b->launches)
        if (uniform(L)) sum_grid += L.grid;
    if (tileL && tile_min * 8 > budget) return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: not enough device memory for the forward scratch of the largest task");
    if (b->slot_stride > 0) fit = (budget - tile_min * 8) / (b->slot_stride * 8);
    if (sum_grid > fit) {
        int64_t others = 0;
        for (auto &L : b->launches) others += uniform(L) ? 1 : 0;
        if (fit < others) return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: not enough device memory for one forward scratch region per kernel class");
        const double shrink = static_cast<double>(fit) / static_cast<double>(sum_grid);
        for (auto &L : b->launches)
            if (uniform(L)) L.grid = std::max(1, static_cast<int>(L.grid * shrink));
    }
    sum_grid = 0;
    for (auto &L : b->launches) {
        if (!uniform(L)) continue;
        L.slot_base = static_cast<int>(sum_grid);
        sum_grid += L.grid;
        if (kClassTab[L.cls].kind == K_GENERIC_GLOBAL) ring_floats = static_cast<int64_t>(L.grid) * 18 * L.wcap;
        max_grid = std::max<int64_t>(max_grid, L.grid);
    }
    // (at least one uniform region: npr_batch_dense runs any task there)
    const int64_t uniform_cells = b->slot_stride * std::max<int64_t>(sum_grid, ntasks ? 1 : 0);
    std::vector<int64_t> region;  // first scratch cell of each workgroup of the launches with their own regions
    int64_t var_total = 0;
    auto own_regions = [&](npr_batch::Launch &L, auto need_of) -> int32_t {
        L.region_first = static_cast<int>(region.size());
        const int64_t room = budget / 8 - uniform_cells - (tileL && &L != tileL ? tile_min : 0);
        int g = 0;
        for (; g < L.grid; ++g) {
            const int64_t need = need_of(rank[L.first + g]);
            if (var_total + need > room) break;
            region.push_back(uniform_cells + var_total);
            var_total += need;
            if (&L == tileL) b->region_end.push_back(uniform_cells + var_total);
        }
        if (g == 0) return NPR_ERR_NOMEM;
        L.grid = g;
        L.slot_base = 0;
        L.own_regions = true;
        max_grid = std::max<int64_t>(max_grid, L.grid);
        return NPR_OK;
    };
    for (auto &L : b->launches) {
        const int kind = kClassTab[L.cls].kind;
        if ((is_one_wave_kind(kind) && b->variable_regions && !uniform(L)) || kind == K_MID) {
            // (k_dp_mid_rs's two sweeps share one set of rows: the forward one stores up to the cut, the backward one above it)
            if (own_regions(L, [&](int32_t g) { return (pad_of[g] + 63) & ~int64_t(63); }) != NPR_OK)
                return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: not enough device memory for the forward scratch of the largest task");
        }
    }
    if (tileL && own_regions(*tileL, [&](int32_t g) { return tile_need[g]; }) != NPR_OK)
        return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: not enough device memory for the forward scratch of the largest task");
    const int64_t tile_total = var_total;
    int64_t own_grid = 0;
    for (auto &L : b->launches) own_grid += L.own_regions ? L.grid : 0;
    const int64_t grid = ntasks ? sum_grid + own_grid : 0;
    if ((e = b->d_tasks.alloc_from(ctx, ntasks)) != hipSuccess || (e = b->d_outs.alloc_from(ctx, ntasks)) != hipSuccess ||
        (e = b->d_queue.alloc_from(ctx, kQueueSlots)) != hipSuccess || (e = b->d_ring.alloc_from(ctx, ring_floats)) != hipSuccess ||
        (e = b->d_region.alloc_from(ctx, region.size())) != hipSuccess ||
        (e = b->d_px.alloc_from(ctx, pair_total)) != hipSuccess ||
        (e = b->d_py.alloc_from(ctx, pair_total)) != hipSuccess || (e = b->d_pp.alloc_from(ctx, pair_total)) != hipSuccess)
        return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: hipMalloc", e);
    b->scratch_cells = static_cast<size_t>(uniform_cells) + static_cast<size_t>(tile_total);
    // The arena only grows, so a batch that fits what is there now goes on without the mutex -- staging the next batch must
    // not wait for the DP pass of the current one, which holds it.  Growing it (or poisoning it) waits for whatever another
    // context's batch is running there.
    if (b->scratch_cells > ctx->arena->cells.load() || poison_byte() >= 0) {
        DeviceArena &ar = *ctx->arena;
        std::lock_guard<std::mutex> lock(ar.mu);
        if (b->scratch_cells > ar.cells) {
            if (ar.F) (void)hipFree(ar.F - DeviceArena::kPad);
            ar.F = nullptr, ar.cells = 0, ++ar.epoch;
            e = hipMalloc(reinterpret_cast<void **>(&ar.F), b->scratch_cells * 8 + 2 * DeviceArena::kPad);
            if (e != hipSuccess && !ctx->cache.empty()) {  // the buffers kept from earlier batches are in the way
                (void)hipGetLastError();
                ctx->cache_flush();
                e = hipMalloc(reinterpret_cast<void **>(&ar.F), b->scratch_cells * 8 + 2 * DeviceArena::kPad);
            }
            if (e != hipSuccess) {
                ar.F = nullptr;
                return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: hipMalloc of the forward scratch", e);
            }
            ar.F += DeviceArena::kPad;
            ar.cells = b->scratch_cells;
        }
        if (poison_byte() >= 0) poison(ar.F, ar.cells * 8), ++ar.epoch;
    }
    tm.lap("hipMalloc");
    if (ntasks) {
        HIP_TRY(ctx, hipMemcpyAsync(b->d_tasks.p, b->tasks.data(), b->d_tasks.bytes(), hipMemcpyHostToDevice, ctx->stream));
        if (!region.empty()) HIP_TRY(ctx, hipMemcpyAsync(b->d_region.p, region.data(), b->d_region.bytes(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (the sequences are in place too: the stream waited for their copy)
    }
    tm.lap("H2D");
    b->outs.resize(ntasks);
    b->stats.n_reads = n_reads;
    b->stats.n_tasks = ntasks;
    b->stats.cells = total_cells;
    b->stats.diagonals = band_entries;
    b->stats.max_width = max_width;
    b->stats.slots = grid;
    {   // report the class that carries most cells
        int64_t best = -1;
        for (const auto &L : b->launches)
            if (L.cells > best) best = L.cells, b->stats.kernel_variant = is_tile_kind(kClassTab[L.cls].kind) ? 2 : (is_register_class(L.cls) ? 1 : 0);
    }
    b->stats.device_bytes = fixed + static_cast<int64_t>(b->scratch_cells) * 8 + ring_floats * 4;
    if (b->params.mode == NPR_MODE_RESCORE_ORIGINAL) {
        const int32_t rc = rescore_stage(b.get());
        if (rc != NPR_OK) return rc;
    }
    drain.armed = false;
    *out = b.release();
    return NPR_OK;
}

static KernelArgs make_args(npr_batch *b) {
    KernelArgs a{};
    a.tasks = b->d_tasks.p;
    a.outs = b->d_outs.p;
    a.queue = b->d_queue.p;
    a.ntasks = static_cast<int32_t>(b->tasks.size());
    a.models = b->ctx->d_models;
    a.seq = b->d_seq.p;
    a.lo = b->d_lo.p;
    a.n = b->d_n.p;
    a.coff = b->d_coff.p;
    a.ctl = b->d_ctl.p;
    a.stripes = b->d_stripes.p;
    a.rowmask = b->d_rowmask.p;
    a.region = nullptr;  // set per launch (own_regions)
    a.F = b->ctx->arena->F;  // (the caller holds the arena's mutex)
    a.slot_stride = b->slot_stride;
    a.px = b->d_px.p;
    a.py = b->d_py.p;
    a.pp = b->d_pp.p;
    a.threshold = static_cast<float>(b->params.posterior_threshold);
    a.ring = b->d_ring.p;
    return a;
}

int32_t npr_batch_run(npr_batch *b, float *kernel_ms) {
    if (!b) return NPR_ERR_INVALID;
    npr_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (kernel_ms) *kernel_ms = 0.f;
    if (b->tasks.empty()) {
        b->ran = true;
        return NPR_OK;
    }
    std::lock_guard<std::mutex> arena_lock(ctx->arena->mu);  // until the DP pass has finished
    ++ctx->arena->epoch;
    DevBuf<unsigned long long> d_prof;  // NPR_TILE_PROF=1 (bring-up): wait cycles of the stripe kernel's wavefronts
    if (std::getenv("NPR_TILE_PROF")) {
        if (d_prof.alloc(8) != hipSuccess) return fail(ctx, NPR_ERR_NOMEM, "npr_batch_run: hipMalloc");
        HIP_TRY(ctx, hipMemsetAsync(d_prof.p, 0, d_prof.bytes(), ctx->stream));
    }
    HIP_TRY(ctx, hipMemsetAsync(b->d_queue.p, 0, sizeof(int32_t) * kQueueSlots, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    // all classes at once, the smallest first, each on its own stream; the main stream waits for all of them,
    // so ev0 -> ev1 brackets the whole DP pass
    std::vector<const npr_batch::Launch *> order;
    for (const auto &L : b->launches) order.push_back(&L);
    std::sort(order.begin(), order.end(), [](const npr_batch::Launch *x, const npr_batch::Launch *y) { return x->cells < y->cells; });
    if (b->pair_rs)  // (staged for the row-scaled kernels under the models of that moment)
        for (int sl = 0; sl < NPR_MAX_MODELS; ++sl)
            if (ctx->model_set[sl] && !rs_model_ok(ctx->models[sl]))
                return fail(ctx, NPR_ERR_MODEL, "npr_batch_run: a model loaded after the batch was staged grows faster than the row-scaled kernels allow: stage the batch again");
    // the row-scaled kernels leave out the two short-gap switch terms of a cell when no loaded model has such a transition (the
    // shipped ones have none): exact zeros either way (npr_rs.h)
    bool sw = false;
    for (int sl = 0; sl < NPR_MAX_MODELS; ++sl)
        if (ctx->model_set[sl] && (ctx->models[sl].T[1 * 5 + 2] != 0.f || ctx->models[sl].T[2 * 5 + 1] != 0.f)) sw = true;
    const bool flat = !sw && flat_gap_emissions(ctx);
    for (size_t i = 0; i < order.size(); ++i) {
        const npr_batch::Launch &L = *order[i];
        const bool last = i + 1 == order.size();
        hipStream_t s = last ? ctx->stream : ctx->side[i % npr_ctx::kSideStreams];
        if (!last) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev0, 0));
        KernelArgs a = make_args(b);
        a.tasks += L.first;
        a.outs += L.first;
        a.ntasks = L.count;
        a.queue += L.cls;
        a.wcap = L.wcap;
        a.slot_base = L.slot_base;
        a.region = L.own_regions ? b->d_region.p + L.region_first : nullptr;
        a.prof = d_prof.p;
        const KClass &kc = kClassTab[L.cls];
        const int rc = kc.kind == K_MID   ? launch_mid_rs(a, kc.R, L.grid, s, sw, flat)
                       : kc.kind == K_RS    ? launch_rs(a, kc.R, L.grid, s, sw, flat)
                       : kc.kind == K_STAIR ? launch_stair(a, kc.R, L.grid, s)
                       : kc.kind == K_TILE ? launch_tile(a, kc.R, L.wcap, L.grid, s, flat_gap_emissions(ctx))
                       : kc.kind == K_TILE_RS ? launch_tile_cs(a, L.wcap, L.grid, s, sw, flat)
                       : kc.kind == K_WIDE ? launch_wide(a, kc.R, kc.NW, L.grid, s)
                                           : launch_generic(a, L.grid, L.threads, L.lds, false, kc.kind == K_GENERIC_GLOBAL, s);
        if (rc != 0) return fail(ctx, NPR_ERR_HIP, "DP kernel launch", static_cast<hipError_t>(rc));
        if (!last) HIP_TRY(ctx, hipEventRecord(ctx->side_done[i % npr_ctx::kSideStreams], s));
    }
    for (size_t i = 0; i + 1 < order.size(); ++i) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_done[i % npr_ctx::kSideStreams], 0));
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (kernel_ms) HIP_TRY(ctx, hipEventElapsedTime(kernel_ms, ctx->ev0, ctx->ev1));
    if (d_prof.p) {
        unsigned long long pf[8];
        HIP_TRY(ctx, hipMemcpy(pf, d_prof.p, sizeof(pf), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[npr tile prof] wavefront cycles: waiting for a neighbour %.3g, for own stores %.3g, at barriers %.3g, total %.3g (k_dp_tile_cs built with -DNPR_TCS_PROF: neighbour, general step, fast loops, total; stripe set-up %.3g, barriers %.3g, task set-up %.3g, own stores %.3g)\n",
                     (double)pf[0], (double)pf[1], (double)pf[2], (double)pf[3], (double)pf[4], (double)pf[5], (double)pf[6], (double)pf[7]);
    }
    // The row-scaled kernels report the tasks for which one exponent per row may not have been enough (TASK_RERUN,
    // npr_device.h): those run again here, with the per-cell-exponent kernel of their frame class, on the scratch regions the
    // first launch had.  Rare -- a row of the alignment ~110 binary orders below the product of the row's largest forward and
    // backward values: an indel of 70+ bases --, so one more small launch per class at most.
    b->outs.resize(b->tasks.size());
    b->task_rerun.assign(b->tasks.size(), 0);
    for (const auto &L : b->launches) {
        if (kClassTab[L.cls].kind != K_RS && kClassTab[L.cls].kind != K_TILE_RS && kClassTab[L.cls].kind != K_MID) continue;
        HIP_TRY(ctx, hipMemcpy(b->outs.data() + L.first, b->d_outs.p + L.first, sizeof(TaskOut) * L.count, hipMemcpyDeviceToHost));
        std::vector<int32_t> again;
        for (int k = L.first; k < L.first + L.count; ++k)
            if (b->outs[k].status == TASK_RERUN) {
                again.push_back(k);
                if (std::getenv("NPR_TIMING")) std::fprintf(stderr, "[npr] task %d (D %d) runs again; the first pass left in its result: npairs (k_dp_mid_rs: why, 1 nothing at the cut / 2 no total / 3 exponents apart / 4 totals apart / 5 range certificate; k_dp_tile_cs: its certificate value) %d, btot_m (k_dp_mid_rs: total' / total) %.9g, btot_e (k_dp_mid_rs: exponent difference) %d, total %g x 2^%d\n", k, b->tasks[k].D, b->outs[k].npairs, b->outs[k].btot_m, b->outs[k].btot_e, b->outs[k].tot_m, b->outs[k].tot_e);
            }
        if (again.empty()) continue;
        std::vector<Task> sub(again.size());
        for (size_t j = 0; j < again.size(); ++j) sub[j] = b->tasks[again[j]];
        DevBuf<Task> d_sub;
        DevBuf<TaskOut> d_subout;
        if (d_sub.alloc(sub.size()) != hipSuccess || d_subout.alloc(sub.size()) != hipSuccess) return fail(ctx, NPR_ERR_NOMEM, "npr_batch_run: hipMalloc");
        HIP_TRY(ctx, hipMemcpy(d_sub.p, sub.data(), sizeof(Task) * sub.size(), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemsetAsync(b->d_queue.p + L.cls, 0, sizeof(int32_t), ctx->stream));
        KernelArgs a = make_args(b);
        a.tasks = d_sub.p, a.outs = d_subout.p, a.ntasks = static_cast<int32_t>(sub.size());
        a.queue += L.cls;
        a.slot_base = L.slot_base;
        a.region = L.own_regions ? b->d_region.p + L.region_first : nullptr;  // (task j of `again` is no larger than the j-th task of the class)
        const int grid = static_cast<int>(std::min<size_t>(sub.size(), static_cast<size_t>(L.grid)));
        a.wcap = L.wcap;
        const int rc = kClassTab[L.cls].kind == K_TILE_RS ? launch_tile(a, 2, L.wcap, grid, ctx->stream) : launch_stair(a, kClassTab[L.cls].R, grid, ctx->stream);
        if (rc != 0) return fail(ctx, NPR_ERR_HIP, "DP kernel launch (second pass)", static_cast<hipError_t>(rc));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<TaskOut> subout(sub.size());
        HIP_TRY(ctx, hipMemcpy(subout.data(), d_subout.p, sizeof(TaskOut) * sub.size(), hipMemcpyDeviceToHost));
        for (size_t j = 0; j < again.size(); ++j) {
            b->outs[again[j]] = subout[j];
            b->task_rerun[again[j]] = 1;
            HIP_TRY(ctx, hipMemcpy(b->d_outs.p + again[j], &subout[j], sizeof(TaskOut), hipMemcpyHostToDevice));
        }
        if (std::getenv("NPR_TIMING")) std::fprintf(stderr, "[npr] class %d: %zu of %d tasks run again with a per-cell exponent\n", L.cls, again.size(), L.count);
    }
    b->ran = true;
    b->finished = false;
    return NPR_OK;
}

int32_t npr_batch_class_stats(const npr_batch *b, int64_t *tasks, int64_t *cells, int32_t cap) {
    if (!b) return NPR_ERR_INVALID;
    for (int c = 0; c < kClasses && c < cap; ++c) {
        if (tasks) tasks[c] = 0;
        if (cells) cells[c] = 0;
    }
    for (const auto &L : b->launches)
        if (L.cls < cap) {
            if (tasks) tasks[L.cls] = L.count;
            if (cells) cells[L.cls] = L.cells;
        }
    return kClasses;
}

int32_t npr_batch_segment_arith(const npr_batch *b, int64_t *seg_off, int32_t *arith, int64_t cap) {
    if (!b || !seg_off) return NPR_ERR_INVALID;
    std::vector<int8_t> of_task(b->tasks.size(), 0);
    for (const auto &L : b->launches)
        if (kClassTab[L.cls].kind == K_RS || kClassTab[L.cls].kind == K_MID)
            for (int k = L.first; k < L.first + L.count; ++k) of_task[k] = (static_cast<size_t>(k) < b->task_rerun.size() && b->task_rerun[k]) ? 0 : 1;
    int64_t n = 0;
    for (int64_t r = 0; r < b->n_reads; ++r) {
        seg_off[r] = n;
        for (int32_t s2 = 0; s2 < b->read_ntasks[r]; ++s2, ++n)
            if (arith && n < cap) arith[n] = of_task[b->task_of[b->read_first_task[r] + s2]];
    }
    seg_off[b->n_reads] = n;
    return NPR_OK;
}

namespace {

// Posterior pairs of every read to the host: one dense D2H, then per read (host threads) its segments' pairs merged
// and sorted by (x, y).  b->task_dst (prefix of the per-task pair counts) and b->pair_off are already set.
int32_t fetch_pairs(npr_batch *b) {
    npr_ctx *ctx = b->ctx;
    if (b->pairs_ready) return NPR_OK;
    StageTimer tm("fetch_pairs");
    const int64_t ntasks = static_cast<int64_t>(b->tasks.size());
    const std::vector<int64_t> &dst = b->task_dst;
    const int32_t *hx = nullptr, *hy = nullptr;
    const float *hp = nullptr;
    const int64_t total = ntasks ? dst[ntasks] : 0;
    if (total) {
        DevBuf<int64_t> d_dst;
        DevBuf<int32_t> d_cx, d_cy;
        DevBuf<float> d_cp;
        hipError_t e;
        if ((e = d_dst.alloc_from(ctx, ntasks + 1)) != hipSuccess || (e = d_cx.alloc_from(ctx, total)) != hipSuccess ||
            (e = d_cy.alloc_from(ctx, total)) != hipSuccess || (e = d_cp.alloc_from(ctx, total)) != hipSuccess)
            return fail(ctx, NPR_ERR_NOMEM, "npr_batch_finish: hipMalloc", e);
        HIP_TRY(ctx, hipMemcpyAsync(d_dst.p, dst.data(), d_dst.bytes(), hipMemcpyHostToDevice, ctx->stream));
        CompactArgs ca{b->d_tasks.p, b->d_outs.p, d_dst.p, static_cast<int32_t>(ntasks), b->d_px.p, b->d_py.p, b->d_pp.p, d_cx.p, d_cy.p, d_cp.p};
        const int rc = launch_compact(ca, ctx->stream);
        if (rc != 0) return fail(ctx, NPR_ERR_HIP, "k_compact launch", static_cast<hipError_t>(rc));
        const size_t need = static_cast<size_t>(total) * 12;
        if (need > ctx->pin_pairs_bytes) {
            if (ctx->pin_pairs) (void)hipHostFree(ctx->pin_pairs);
            ctx->pin_pairs = nullptr, ctx->pin_pairs_bytes = 0;
            if ((e = hipHostMalloc(&ctx->pin_pairs, need + need / 4, hipHostMallocDefault)) != hipSuccess)
                return fail(ctx, NPR_ERR_NOMEM, "npr_batch_finish: hipHostMalloc", e);
            ctx->pin_pairs_bytes = need + need / 4;
        }
        int32_t *px_h = static_cast<int32_t *>(ctx->pin_pairs), *py_h = px_h + total;
        float *pp_h = reinterpret_cast<float *>(py_h + total);
        HIP_TRY(ctx, hipMemcpyAsync(px_h, d_cx.p, d_cx.bytes(), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(py_h, d_cy.p, d_cy.bytes(), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(pp_h, d_cp.p, d_cp.bytes(), hipMemcpyDeviceToHost, ctx->stream));
        hx = px_h, hy = py_h, hp = pp_h;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    tm.lap("compact + D2H");
    const int64_t n = b->n_reads;
    b->pairs.resize(b->pair_off[n]);
    parallel_for(n, ctx->host_threads, [&](int64_t i) {
        if (b->read_status[i] != NPR_OK) return;
        Pair *pp = b->pairs.data() + b->pair_off[i];
        int64_t c = 0;
        for (int32_t s = 0; s < b->read_ntasks[i]; ++s) {
            const int32_t k = b->task_of[b->read_first_task[i] + s];
            for (int64_t q = dst[k]; q < dst[k + 1]; ++q) pp[c++] = Pair{hx[q], hy[q], hp[q]};
        }
        // order by (x, y).  The pairs of a read number about two per reference base, so when the reference span is
        // not much longer than the list a counting sort on x (+ insertion sort of the few pairs sharing an x) beats
        // a comparison sort several times over; chained records that span a whole contig keep std::sort.
        const int64_t span = b->ref_len[i];
        if (c > 64 && span <= 4 * c) {
            thread_local std::vector<int32_t> start;
            thread_local std::vector<Pair> tmp;
            start.assign(span + 2, 0);
            bool ok = true;
            for (int64_t q = 0; q < c; ++q) {
                if (pp[q].x < 0 || pp[q].x >= span) {
                    ok = false;
                    break;
                }
                ++start[pp[q].x + 1];
            }
            if (ok) {
                for (int64_t x = 0; x < span; ++x) start[x + 1] += start[x];
                tmp.resize(c);
                for (int64_t q = 0; q < c; ++q) tmp[start[pp[q].x]++] = pp[q];  // start[x] is now the END of group x
                int64_t g = 0;
                for (int64_t q = 0; q < c; ++q) {  // insertion sort inside each x-group
                    if (q > 0 && tmp[q].x != tmp[q - 1].x) g = q;
                    Pair v = tmp[q];
                    int64_t k = q;
                    while (k > g && tmp[k - 1].y > v.y) tmp[k] = tmp[k - 1], --k;
                    tmp[k] = v;
                }
                std::copy(tmp.begin(), tmp.end(), pp);
            } else {
                std::sort(pp, pp + c, [](const Pair &a, const Pair &d) { return a.x != d.x ? a.x < d.x : a.y < d.y; });
            }
        } else {
            std::sort(pp, pp + c, [](const Pair &a, const Pair &d) { return a.x != d.x ? a.x < d.x : a.y < d.y; });
        }
    });
    tm.lap("merge + sort");
    b->pairs_ready = true;
    return NPR_OK;
}

// NPR_MODE_RESCORE_ORIGINAL on the device (npr_stats.hip k_rescore_table / k_rescore_sum; the reference's call site: alignmentUncertainty.py:41,
// the analysis that runs on every experiment by default, pipeline.py:81).  At staging the guide's M runs go up once (12 bytes per run) and are
// spread into a table over the reference positions of each read's window; every pass then is one sweep over the pairs where the DP kernels
// left them and eight bytes per read coming back -- no pair crosses PCIe, and the guide's operations are not copied until somebody asks for
// the cigars.  rescore_stage leaves b->rs_staged false when the fixed-point sum could not be exact (a threshold below 2^-20, a guide of
// 2^(53 - shift) M columns): the host stage scores then.
int32_t rescore_stage(npr_batch *b) {
    npr_ctx *ctx = b->ctx;
    const int64_t n = b->n_reads;
    StageTimer tm("rescore_stage");
    b->rs_staged = false;
    b->rs_columns.assign(n, 0), b->rs_kept.assign(n, 0);
    std::vector<int64_t> run_off(n + 1, 0), gx_off(n + 1, 0);
    parallel_for(n, ctx->host_threads, [&](int64_t i) {
        int64_t runs = 0, cols = 0, kept = 0;
        for (int64_t q = b->guide_off[i]; q < b->guide_off[i + 1]; ++q) {
            const int32_t len = b->guide_ops[2 * q + 1];
            kept += len > 0;
            if (b->guide_ops[2 * q] == NPR_OP_M && len > 0) ++runs, cols += len;
        }
        b->rs_columns[i] = cols, b->rs_kept[i] = kept, run_off[i + 1] = b->read_status[i] == NPR_OK ? runs : 0;
    });
    if (ctx->opt[NPR_OPT_HOST_MEA] != 0 || n == 0) return NPR_OK;
    int e2 = 0;
    (void)std::frexp(b->params.posterior_threshold, &e2);  // threshold = m * 2^e2, m in [0.5, 1): an fp32 p >= threshold is a multiple of 2^(e2 - 1 - 23)
    const int shift = 24 - e2;
    if (!(b->params.posterior_threshold > 0.0) || shift > 44 || shift < 0) return NPR_OK;
    for (int64_t i = 0; i < n; ++i) {
        if (b->rs_columns[i] >= (int64_t(1) << (53 - shift))) return NPR_OK;
        run_off[i + 1] += run_off[i];
        gx_off[i + 1] = gx_off[i] + (b->read_status[i] == NPR_OK ? b->ref_len[i] + 1 : 0);
    }
    // the runs through the context's pinned staging buffer when it is there (157 MB for 8192 reads of 8 kb: pageable memory halves the copy's rate)
    std::vector<int32_t> runs_v;
    int32_t *runs = nullptr;
    const size_t run_bytes = sizeof(int32_t) * 3 * static_cast<size_t>(run_off[n]);
    if (ctx->pin_stage && ctx->pin_stage_bytes >= run_bytes) runs = static_cast<int32_t *>(ctx->pin_stage);
    else runs_v.resize(3 * static_cast<size_t>(run_off[n])), runs = runs_v.data();
    parallel_for(n, ctx->host_threads, [&](int64_t i) {
        if (b->read_status[i] != NPR_OK) return;
        int32_t *out = runs + 3 * run_off[i];
        int64_t x = 0, y = 0;
        for (int64_t q = b->guide_off[i]; q < b->guide_off[i + 1]; ++q) {
            const int32_t op = b->guide_ops[2 * q], len = b->guide_ops[2 * q + 1];
            if (op == NPR_OP_M) {
                if (len > 0) out[0] = static_cast<int32_t>(x), out[1] = static_cast<int32_t>(y), out[2] = len, out += 3;
                x += len, y += len;
            } else if (op == NPR_OP_I) {
                y += len;
            } else {
                x += len;
            }
        }
    });
    tm.lap("runs");
    DevBuf<int64_t> d_run_off;
    DevBuf<int32_t> d_runs;
    hipError_t e;
    if ((e = d_run_off.alloc_from(ctx, n + 1)) != hipSuccess || (e = b->d_rs_gx_off.alloc_from(ctx, n + 1)) != hipSuccess ||
        (e = d_runs.alloc_from(ctx, std::max<size_t>(3 * static_cast<size_t>(run_off[n]), 1))) != hipSuccess ||
        (e = b->d_rs_gy.alloc_from(ctx, std::max<int64_t>(gx_off[n], 1))) != hipSuccess)
        return fail(ctx, NPR_ERR_NOMEM, "npr_batch_create: hipMalloc (rescore tables)", e);
    HIP_TRY(ctx, hipMemcpyAsync(d_run_off.p, run_off.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(b->d_rs_gx_off.p, gx_off.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    if (run_bytes) HIP_TRY(ctx, hipMemcpyAsync(d_runs.p, runs, run_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(b->d_rs_gy.p, 0xff, sizeof(int32_t) * std::max<int64_t>(gx_off[n], 1), ctx->stream));
    RescoreArgs ra{static_cast<int32_t>(n), 0, d_run_off.p, d_runs.p, b->d_rs_gx_off.p, b->d_rs_gy.p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, shift};
    const int rc = launch_rescore_table(ra, ctx->stream);
    if (rc != 0) return fail(ctx, NPR_ERR_HIP, "k_rescore_table launch", static_cast<hipError_t>(rc));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (the staging buffer and d_runs go back)
    tm.lap("table");
    b->rs_shift = shift, b->rs_staged = true;
    return NPR_OK;
}

int32_t rescore_sum(npr_batch *b, std::vector<double> &score) {
    npr_ctx *ctx = b->ctx;
    const int64_t n = b->n_reads, ntasks = static_cast<int64_t>(b->tasks.size());
    DevBuf<unsigned long long> d_sum;
    hipError_t e;
    if ((e = d_sum.alloc_from(ctx, n)) != hipSuccess) return fail(ctx, NPR_ERR_NOMEM, "npr_batch_finish: hipMalloc (rescore sums)", e);
    HIP_TRY(ctx, hipMemsetAsync(d_sum.p, 0, sizeof(unsigned long long) * n, ctx->stream));
    RescoreArgs ra{static_cast<int32_t>(n), static_cast<int32_t>(ntasks), nullptr, nullptr, b->d_rs_gx_off.p, b->d_rs_gy.p, b->d_tasks.p, b->d_outs.p,
                   b->d_px.p, b->d_py.p, b->d_pp.p, d_sum.p, b->rs_shift};
    const int rc = launch_rescore_sum(ra, ctx->stream);
    if (rc != 0) return fail(ctx, NPR_ERR_HIP, "k_rescore_sum launch", static_cast<hipError_t>(rc));
    std::vector<unsigned long long> sum(n);
    HIP_TRY(ctx, hipMemcpyAsync(sum.data(), d_sum.p, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    score.assign(n, 0.0);
    for (int64_t i = 0; i < n; ++i)
        if (b->rs_columns[i] > 0) score[i] = std::ldexp(static_cast<double>(sum[i]), -b->rs_shift) / static_cast<double>(b->rs_columns[i]);
    return NPR_OK;
}

// MEA chain + cigar of every read on the device (npr_mea.hip): only the ops cross PCIe.  Returns 1 when some read
// needs the host stage instead (a chain reaching back further than the prefix-maximum ring), NPR_OK or an error.
int32_t device_mea(npr_batch *b) {
    npr_ctx *ctx = b->ctx;
    // the tables are carved out of the arena when they fit -- unless the context runs next to others (NPR_OPT_OVERLAP): then
    // they live in buffers of its own and the stage need not wait for another batch's DP pass
    std::unique_lock<std::mutex> arena_lock(ctx->arena->mu, std::defer_lock);
    if (!ctx->overlap) arena_lock.lock();
    ++ctx->arena->epoch;
    StageTimer tm("device_mea");
    const int64_t n = b->n_reads, ntasks = static_cast<int64_t>(b->tasks.size());
    std::vector<int64_t> rx(n + 1, 0), ry(n + 1, 0), rp(n + 1, 0), ot(n + 1, 0), od(n + 1, 0);
    for (int64_t i = 0; i < n; ++i) {  // a read that already failed gets empty tables: its pairs are skipped as out of range
        const bool ok = b->results[i].status == NPR_OK;
        const int64_t lX = ok ? b->ref_len[i] : 0, lY = ok ? b->read_len[i] : 0, np = ok ? b->pair_off[i + 1] - b->pair_off[i] : 0;
        rx[i + 1] = rx[i] + lX + 1;
        ry[i + 1] = ry[i] + lY;
        rp[i + 1] = rp[i] + np;
        ot[i + 1] = ot[i] + 3 * std::min({np, lX, lY}) + 2;  // (D, I, M) per chain pair, one trailing (D, I)
    }
    // the LDS-ring kernel takes the few reads the register window gives up on: as many read positions as the LDS
    // holds with one workgroup per CU; a read whose pairs reach back further than that is reported and the batch takes
    // the host stage
    const int ring = 8192;
    const int64_t total = rp[n];
    // the pieces the chain of every read is cut into (npr_mea.hip k_mea_cuts): about 2000 posterior pairs (1200 kept) each
    // ... fewer in a small batch, so that the pieces (one lane each, a serial walk) still fill the chip: 1000 reads of 1 kb as 1000
    // pieces of 1100 kept pairs took 0.9 ms where 14 000 pieces of 80 take 0.1
    constexpr int64_t kMaxPieces = 64, kLanesWanted = 64 * 5 * 256;
    const int64_t kPiecePairs = std::min<int64_t>(2048, std::max<int64_t>(128, total / kLanesWanted));
    std::vector<int32_t> np(n);
    int64_t n_pieces = 0;
    for (int64_t i = 0; i < n; ++i) np[i] = static_cast<int32_t>(std::min(kMaxPieces, std::max<int64_t>(1, (rp[i + 1] - rp[i] + kPiecePairs - 1) / kPiecePairs))), n_pieces += np[i];
    if (!ctx->mea) ctx->mea = new MeaScratch;
    MeaScratch &m = *ctx->mea;
    hipError_t e;
    // per-position tables of one read in LDS (count + scan + scatter in one kernel) when the longest span fits
    // ... read by read (round 4: one read of more than 16 k bases used to send its whole batch through the global-memory kernels)
    const int64_t lds_span = ctx->opt[NPR_OPT_MEA_GLOBAL_SORT] != 0 ? 0 : 16 * 1024;
    int64_t span = 0;  // the widest table among the reads that sort in LDS
    std::vector<int64_t> cnt_off(n + 1, -1);
    int64_t cnt_total = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t sp = std::max(rx[i + 1] - rx[i], ry[i + 1] - ry[i]);
        if (sp <= lds_span) span = std::max(span, sp);
        else cnt_off[i] = cnt_total, cnt_total += rx[i + 1] - rx[i];
    }
    const bool sort_in_lds = cnt_total == 0;
    const size_t ntask_map = b->task_of.size();
    // The forward scratch of the DP launches is idle now and usually far larger than what this stage needs: carve the
    // tables out of it (a batch that fills the device's memory leaves nothing to hipMalloc).  Else: grow-only buffers.
    const size_t n_cnt = sort_in_lds ? 1 : static_cast<size_t>(cnt_total);
    {
        auto al = [](size_t bytes) { return (bytes + 255) & ~size_t(255); };
        const size_t need = al(8 * 5 * (n + 1)) + al(8 * n) + al(8 * (n + 1)) + 2 * al(4 * n_cnt) + al(4 * (ry[n] + 1)) + al(4 * (12 * total + 16)) +
                            al(4 * 6 * n) + al(4 * 2 * ot[n]) + al(4 * (3 * n + ntask_map)) + al(4 * ot[n]) + al(4 * (4 * n_pieces + 4 * n));
        const bool arena_fits = ctx->arena->F && need <= static_cast<size_t>(ctx->arena->cells.load()) * 8 && ctx->opt[NPR_OPT_MEA_OWN_SCRATCH] == 0;
        bool in_arena = !ctx->overlap && arena_fits;
        for (;;) {
            char *cur = ctx->arena->F;
            if (in_arena && poison_byte() >= 0) {  // the DP launches are done (their streams feed this one): the tables start from poison
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                poison(ctx->arena->F, need);
            }
            auto take = [&](auto &buf, size_t count) -> hipError_t {
                using T = std::remove_pointer_t<decltype(buf.p)>;
                if (!in_arena) return buf.reserve(count);
                buf.borrow(reinterpret_cast<T *>(cur), count);
                cur += al(sizeof(T) * count);
                return hipSuccess;
            };
            if ((e = take(m.off, 5 * (n + 1))) == hipSuccess && (e = take(m.mass, n)) == hipSuccess && (e = take(m.od, n + 1)) == hipSuccess &&
                (e = take(m.cnt, n_cnt)) == hipSuccess && (e = take(m.start, n_cnt)) == hipSuccess && (e = take(m.col, ry[n] + 1)) == hipSuccess &&
                (e = take(m.sorted, 12 * total + 16)) == hipSuccess && (e = take(m.small, 6 * n)) == hipSuccess && (e = take(m.tmp, 2 * ot[n])) == hipSuccess &&
                (e = take(m.map, 3 * n + ntask_map)) == hipSuccess && (e = take(m.dense, ot[n])) == hipSuccess &&
                (e = take(m.pieces, 4 * n_pieces + 4 * n)) == hipSuccess)
                break;
            (void)hipGetLastError();
            if (!in_arena && ctx->overlap && arena_fits) {
                // A pipelined job's context keeps these tables in buffers of its own (NPR_OPT_OVERLAP) so that it need not wait for the batch
                // that is running in the device's shared scratch -- when they do not fit beside the batches in flight (long reads: 48 bytes per
                // pair, three chunks on the device) it waits after all, and gives back what it had reserved.
                m.off.release(), m.mass.release(), m.od.release(), m.cnt.release(), m.start.release(), m.col.release(), m.sorted.release();
                m.small.release(), m.tmp.release(), m.map.release(), m.dense.release(), m.pieces.release();
                ctx->cache_flush();
                arena_lock.lock();
                // (`arena_fits` was read before the lock: another context may have released or regrown the shared scratch since)
                if (!(ctx->arena->F && need <= static_cast<size_t>(ctx->arena->cells.load()) * 8)) return 1;
                ++ctx->arena->epoch;
                in_arena = true;
                continue;
            }
            return 1;  // no room on the device: the host stage takes the batch
        }
    }
    HIP_TRY(ctx, hipMemcpyAsync(m.map.p, b->read_first_task.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(m.map.p + n, b->read_ntasks.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(m.map.p + 2 * n, b->task_of.data(), sizeof(int32_t) * ntask_map, hipMemcpyHostToDevice, ctx->stream));
    std::vector<int32_t> order(n);  // longest first: the per-read kernels end together instead of waiting for a late long read
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return rp[x + 1] - rp[x] > rp[y + 1] - rp[y]; });
    HIP_TRY(ctx, hipMemcpyAsync(m.map.p + 2 * n + ntask_map, order.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    {
        // layout of m.pieces: np[n] | poff[n] | pboff[n] | lane_read[P] | lane_piece[P] | pbest[P] | pb[P + n]; lanes in the reads' order
        std::vector<int32_t> tab(3 * n + 2 * n_pieces);
        int32_t *const t_np = tab.data(), *const t_poff = t_np + n, *const t_pboff = t_poff + n, *const t_lr = t_pboff + n, *const t_lp = t_lr + n_pieces;
        int64_t at = 0;
        for (int64_t k = 0; k < n; ++k) {
            const int32_t r = order[k];
            t_np[r] = np[r], t_poff[r] = static_cast<int32_t>(at), t_pboff[r] = static_cast<int32_t>(at + k);
            for (int32_t j = 0; j < np[r]; ++j) t_lr[at + j] = r, t_lp[at + j] = j;
            at += np[r];
        }
        HIP_TRY(ctx, hipMemcpyAsync(m.pieces.p, tab.data(), sizeof(int32_t) * tab.size(), hipMemcpyHostToDevice, ctx->stream));
    }
    std::vector<int64_t> offs(5 * (n + 1));
    std::copy(cnt_off.begin(), cnt_off.end(), offs.begin() + 4 * (n + 1));
    std::copy(rx.begin(), rx.end(), offs.begin());
    std::copy(ry.begin(), ry.end(), offs.begin() + (n + 1));
    std::copy(rp.begin(), rp.end(), offs.begin() + 2 * (n + 1));
    std::copy(ot.begin(), ot.end(), offs.begin() + 3 * (n + 1));
    HIP_TRY(ctx, hipMemcpyAsync(m.off.p, offs.data(), m.off.bytes(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(m.small.p, 0, m.small.bytes(), ctx->stream));
    MeaArgs a{};
    a.tasks = b->d_tasks.p, a.outs = b->d_outs.p, a.ntasks = static_cast<int32_t>(ntasks), a.n_reads = static_cast<int32_t>(n);
    a.px = b->d_px.p, a.py = b->d_py.p, a.pp = b->d_pp.p;
    a.rx_off = m.off.p, a.ry_off = m.off.p + (n + 1), a.rp_off = m.off.p + 2 * (n + 1), a.ot_off = m.off.p + 3 * (n + 1);
    a.cnt = m.cnt.p, a.start = m.start.p, a.colsum = m.col.p;
    a.sx = m.sorted.p, a.sy = m.sorted.p + total + 1, a.sq = m.sorted.p + 2 * (total + 1), a.back = m.sorted.p + 3 * (total + 1);
    a.kx = m.sorted.p + 4 * (total + 1), a.ky = m.sorted.p + 5 * (total + 1), a.kq = m.sorted.p + 6 * (total + 1), a.kback = m.sorted.p + 7 * (total + 1);
    a.vrec = reinterpret_cast<int4 *>(m.sorted.p + ((8 * (total + 1) + 3) & ~int64_t(3)));  // (16-byte records: the arena's tables start 256-byte aligned)
    a.best_who = m.small.p, a.read_flag = m.small.p + n, a.n_ops = m.small.p + 2 * n, a.chain_len = m.small.p + 3 * n, a.kept = m.small.p + 4 * n, a.max_run = m.small.p + 5 * n;
    a.chain_mass = m.mass.p;
    a.np = m.pieces.p, a.poff = m.pieces.p + n, a.pboff = m.pieces.p + 2 * n, a.lane_read = m.pieces.p + 3 * n, a.lane_piece = m.pieces.p + 3 * n + n_pieces;
    a.pbest = m.pieces.p + 3 * n + 2 * n_pieces, a.pb = m.pieces.p + 3 * n + 3 * n_pieces, a.n_pieces = static_cast<int32_t>(n_pieces);
    a.gap_gamma = b->params.gap_gamma, a.match_gamma = b->params.match_gamma, a.ring = ring;
    a.ring_only = ctx->opt[NPR_OPT_MEA_RING_ONLY] != 0 ? 1 : 0;
    a.read_first = m.map.p, a.read_ntasks = m.map.p + n, a.task_of = m.map.p + 2 * n, a.order = m.map.p + 2 * n + ntask_map;
    a.sort_lds_bytes = static_cast<int32_t>(4 * span);
    a.sort_threads = ctx->overlap == 1 ? 512 : 0;  // (beside a DP pass: workgroups that fit the half it leaves -- 1024 threads: the job 388 ms instead of 353, 256: 361)  // (beside a DP pass that leaves part of every SIMD: a workgroup that fits there)
    a.any_global_sort = sort_in_lds ? 0 : 1;
    a.cnt_off = m.off.p + 4 * (n + 1);
    a.ops_tmp = m.tmp.p, a.od_off = m.od.p;
    int rc = launch_mea_sort(a, ctx->stream);
    if (rc == 0) rc = launch_mea_chain(a, ctx->stream);
    if (rc != 0) return fail(ctx, NPR_ERR_HIP, "MEA kernel launch", static_cast<hipError_t>(rc));
    std::vector<int32_t> small(6 * n);
    std::vector<int64_t> mass(n);
    HIP_TRY(ctx, hipMemcpyAsync(small.data(), m.small.p, m.small.bytes(), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(mass.data(), m.mass.p, m.mass.bytes(), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    tm.lap("sort + chain + trace");
    const int32_t *flag = small.data() + n, *nops = small.data() + 2 * n, *clen = small.data() + 3 * n;
    int32_t longest = 0;  // run of the batch's cigars
    for (int64_t i = 0; i < n; ++i)
        if (b->results[i].status == NPR_OK && flag[i] == NPR_ERR_CAPACITY) return 1;
    for (int64_t i = 0; i < n; ++i) {
        npr_read_result &r = b->results[i];
        if (r.status == NPR_OK && flag[i] != 0) r.status = flag[i];
        const int64_t k = r.status == NPR_OK ? nops[i] : 0;
        od[i + 1] = od[i] + k;
        if (k) longest = std::max(longest, small[5 * n + i]);
        r.n_ops = k;
        r.score = (r.status == NPR_OK && clen[i] > 0) ? static_cast<double>(mass[i]) / (static_cast<double>(clen[i]) * PROB_ONE) : 0.0;
    }
    b->ops_off = od;
    b->ops_words = 2 * od[n];
    b->have_pairs_form = false, b->have_packed_form = true;
    if (od[n] > b->packed_cap)  // kept when the batch is finished again; else one a destroyed batch left behind, if it is large enough
        for (size_t i = 0; i < ctx->packed_pool.size(); ++i)
            if (ctx->packed_pool[i].cap >= od[n]) {
                b->packed = std::move(ctx->packed_pool[i].p), b->packed_cap = ctx->packed_pool[i].cap;
                ctx->packed_pool.erase(ctx->packed_pool.begin() + static_cast<std::ptrdiff_t>(i));
                break;
            }
    if (od[n] > b->packed_cap) {
        b->packed.reset(new uint32_t[od[n] + od[n] / 8]);  // (some room: the chunks of a job are about the same size, not exactly)
        b->packed_cap = od[n] + od[n] / 8;
    }
    if (od[n]) {
        // One packed word per op (length << 2 | op), through the pinned staging in pieces: the host threads move a piece into the
        // batch's buffer while the next ones cross.  When no run of the batch is longer than 14 bits (a deletion of 16 k bases: the rule)
        // the words cross as their low halves, 147 MB instead of 295 for the bench's 24576 reads, and the move widens them.
        const bool narrow = longest < (1 << 14) && ctx->opt[NPR_OPT_MEA_WIDE_OPS] == 0 &&
                            sizeof(uint16_t) * static_cast<size_t>(od[n]) <= m.sorted.bytes();  // (the sorted pairs are done with)
        a.ops_dense = m.dense.p;  // (sized for the bound ot[n] >= od[n])
        a.ops_dense16 = narrow ? reinterpret_cast<uint16_t *>(m.sorted.p) : nullptr;
        HIP_TRY(ctx, hipMemcpyAsync(m.od.p, od.data(), m.od.bytes(), hipMemcpyHostToDevice, ctx->stream));
        if ((rc = launch_mea_gather(a, ctx->stream)) != 0) return fail(ctx, NPR_ERR_HIP, "k_mea_gather launch", static_cast<hipError_t>(rc));
        const size_t word = narrow ? sizeof(uint16_t) : sizeof(uint32_t), need = word * static_cast<size_t>(od[n]);
        if (need > ctx->pin_pairs_bytes) {
            if (ctx->pin_pairs) (void)hipHostFree(ctx->pin_pairs);
            ctx->pin_pairs = nullptr, ctx->pin_pairs_bytes = 0;
            if ((e = hipHostMalloc(&ctx->pin_pairs, need + need / 4, hipHostMallocDefault)) != hipSuccess)
                return fail(ctx, NPR_ERR_NOMEM, "npr_batch_finish: hipHostMalloc", e);
            ctx->pin_pairs_bytes = need + need / 4;
        }
        constexpr int64_t kOpsPieces = 48;
        const int64_t nops_all = od[n], pieces = std::min<int64_t>(kOpsPieces, (nops_all + (1 << 20) - 1) >> 20);
        const int64_t piece = ((nops_all + pieces - 1) / pieces + 63) & ~int64_t(63);
        while (static_cast<int64_t>(ctx->ops_events.size()) < pieces) {
            hipEvent_t ev;
            if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return fail(ctx, NPR_ERR_HIP, "hipEventCreate", e);
            ctx->ops_events.push_back(ev);
        }
        const char *dev = narrow ? reinterpret_cast<const char *>(a.ops_dense16) : reinterpret_cast<const char *>(m.dense.p);
        char *pin = static_cast<char *>(ctx->pin_pairs);
        for (int64_t c = 0; c < pieces; ++c) {
            const int64_t lo = std::min(nops_all, c * piece), hi = std::min(nops_all, lo + piece);
            if (hi > lo) HIP_TRY(ctx, hipMemcpyAsync(pin + word * lo, dev + word * lo, word * static_cast<size_t>(hi - lo), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipEventRecord(ctx->ops_events[c], ctx->stream));
        }
        uint32_t *out = b->packed.get();
        std::atomic<int> failed{0};
        parallel_for(pieces, ctx->host_threads, [&](int64_t c) {  // (the items are handed out in order)
            if (hipSetDevice(ctx->device) != hipSuccess || hipEventSynchronize(ctx->ops_events[c]) != hipSuccess) {  // (a worker thread starts on device 0)
                failed = 1;
                return;
            }
            const int64_t lo = std::min(nops_all, c * piece), hi = std::min(nops_all, lo + piece);
            if (narrow) {
                const uint16_t *src = reinterpret_cast<const uint16_t *>(pin);
                for (int64_t i = lo; i < hi; ++i) out[i] = src[i];
            } else {
                std::memcpy(out + lo, pin + word * lo, word * static_cast<size_t>(hi - lo));
            }
        });
        if (failed) return fail(ctx, NPR_ERR_HIP, "npr_batch_finish: D2H of the ops", hipGetLastError());
    }
    tm.lap("gather + D2H of the ops");
    if (od[n]) b->dev_ops = m.dense.p, b->dev_od = m.od.p, b->dev_ops_epoch = ctx->arena->epoch;
    return NPR_OK;
}

}  // namespace

static int32_t batch_finish_impl(npr_batch *b);

int32_t npr_batch_finish(npr_batch *b) {
    try {
        return batch_finish_impl(b);
    } catch (const std::exception &) {
        return fail(b ? b->ctx : nullptr, NPR_ERR_NOMEM, "npr_batch_finish: out of host memory");
    }
}

static int32_t batch_finish_impl(npr_batch *b) {
    if (!b) return NPR_ERR_INVALID;
    npr_ctx *ctx = b->ctx;
    if (!b->ran) return fail(ctx, NPR_ERR_STATE, "npr_batch_finish before npr_batch_run");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    StageTimer tm("batch_finish");
    const int64_t ntasks = static_cast<int64_t>(b->tasks.size());
    const int64_t n = b->n_reads;
    std::vector<int64_t> &dst = b->task_dst;
    dst.assign(ntasks + 1, 0);
    if (ntasks) HIP_TRY(ctx, hipMemcpy(b->outs.data(), b->d_outs.p, b->d_outs.bytes(), hipMemcpyDeviceToHost));
    for (int64_t k = 0; k < ntasks; ++k) dst[k + 1] = dst[k] + std::min(b->outs[k].npairs, b->tasks[k].pair_cap);
    b->results.assign(n, npr_read_result{});
    b->pair_off.assign(n + 1, 0);
    b->pairs_ready = false;
    const double LN2 = 0.69314718055994530942;
    for (int64_t i = 0; i < n; ++i) {
        npr_read_result &r = b->results[i];
        r.status = b->read_status[i];
        r.n_segments = b->read_ntasks[i];
        int64_t c = 0;
        if (r.status == NPR_OK)
            for (int32_t s = 0; s < b->read_ntasks[i]; ++s) {
                const int32_t k = b->task_of[b->read_first_task[i] + s];
                const TaskOut &o = b->outs[k];
                if (o.status != NPR_OK && r.status == NPR_OK) r.status = o.status;
                r.cells += b->task_cells[k];
                if (o.tot_m > 0.f) r.loglik += (std::log2(static_cast<double>(o.tot_m)) + o.tot_e) * LN2;
                if (o.btot_m > 0.f) r.loglik_bwd += (std::log2(static_cast<double>(o.btot_m)) + o.btot_e) * LN2;
                c += dst[k + 1] - dst[k];
            }
        r.n_pairs = c;
        b->pair_off[i + 1] = b->pair_off[i] + c;
    }
    tm.lap("task results");
    // --- rescore mode: the guide's M columns looked up where the pairs lie (round 5) ---
    std::vector<double> dev_score;
    bool have_dev_score = false;
    if (b->params.mode == NPR_MODE_RESCORE_ORIGINAL && n > 0 && ntasks > 0 && b->rs_staged && ctx->opt[NPR_OPT_HOST_MEA] == 0) {
        const int32_t rc = rescore_sum(b, dev_score);
        if (rc < 0) return rc;
        have_dev_score = true;
        tm.lap("device rescore");
    }
    // --- realign and all-posteriors modes: chain and cigar on the device, the pairs stay in HBM until npr_batch_pairs asks for them ---
    if ((b->params.mode == NPR_MODE_REALIGN || b->params.mode == NPR_MODE_ALL_POSTERIORS) && n > 0 && ntasks > 0 && ctx->opt[NPR_OPT_HOST_MEA] == 0) {
        int64_t scratch = 0;
        for (int64_t i = 0; i < n; ++i) scratch += 8 * (b->ref_len[i] + 1) + 4 * b->read_len[i] + 36 * std::min(b->ref_len[i], b->read_len[i]) + 128;
        scratch += 48 * b->pair_off[n];
        size_t mem_free = 0, mem_total = 0;
        const size_t arena_bytes = ctx->arena->cells.load() * 8;
        if (static_cast<size_t>(scratch) <= arena_bytes ||
            (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && static_cast<size_t>(scratch) < mem_free / 2)) {
            const int32_t rc = device_mea(b);
            if (rc < 0) return rc;
            if (rc == NPR_OK) {
                tm.lap("device MEA");
                b->finished = true;
                return NPR_OK;
            }
        }
    }
    // --- host stage: what the device stages could not take (per-position tables that would not fit: records chained across a whole contig;
    // a fixed-point sum that could not be exact), and NPR_OPT_HOST_MEA ---
    if (!have_dev_score) {
        const int32_t rc = fetch_pairs(b);
        if (rc != NPR_OK) return rc;
    }
    if (b->params.mode == NPR_MODE_RESCORE_ORIGINAL) {
        // --rescoreOriginalAlignment: ops verbatim (alignmentUncertainty.py:51-52), new score.  The guide's operations are not copied here
        // (10^7-10^8 per batch): npr_batch_ops / npr_batch_ops_packed make the form they are asked for from b->guide_ops
        b->ops_off.assign(n + 1, 0);
        parallel_for(n, ctx->host_threads, [&](int64_t i) {
            npr_read_result &r = b->results[i];
            if (r.status != NPR_OK) return;
            r.n_ops = b->rs_kept[i], b->ops_off[i + 1] = b->rs_kept[i];
            r.score = have_dev_score ? dev_score[i]
                                     : rescore(b->guide_ops.data() + 2 * b->guide_off[i], b->guide_off[i + 1] - b->guide_off[i], b->pairs.data() + b->pair_off[i], r.n_pairs);
        });
        for (int64_t i = 0; i < n; ++i) b->ops_off[i + 1] += b->ops_off[i];
        b->ops_words = 2 * b->ops_off[n];
        b->ops_from_guide = true, b->have_pairs_form = false, b->have_packed_form = false;
        tm.lap("scores");
        b->finished = true;
        return NPR_OK;
    }
    std::vector<std::vector<int32_t>> per_read_ops(n);
    parallel_for(n, ctx->host_threads, [&](int64_t i) {
        npr_read_result &r = b->results[i];
        if (r.status != NPR_OK) return;
        const int32_t rc = mea_cigar(b->ref_len[i], b->read_len[i], b->pairs.data() + b->pair_off[i], r.n_pairs, b->params.gap_gamma, b->params.match_gamma, per_read_ops[i], r.score);
        if (rc != NPR_OK) r.status = rc;
        r.n_ops = static_cast<int64_t>(per_read_ops[i].size() / 2);
    });
    tm.lap("MEA + cigar");
    b->ops_off.assign(n + 1, 0);
    for (int64_t i = 0; i < n; ++i) b->ops_off[i + 1] = b->ops_off[i] + static_cast<int64_t>(per_read_ops[i].size() / 2);
    b->ops_words = 2 * b->ops_off[n];
    if (b->ops_words > b->ops_cap) {
        b->ops.reset(new int32_t[b->ops_words]);
        b->ops_cap = b->ops_words;
    }
    for (int64_t i = 0; i < n; ++i) std::copy(per_read_ops[i].begin(), per_read_ops[i].end(), b->ops.get() + 2 * b->ops_off[i]);
    b->have_pairs_form = true, b->have_packed_form = false;
    tm.lap("gather ops");
    b->finished = true;
    return NPR_OK;
}

void npr_batch_destroy(npr_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    if (b->packed && b->ctx->packed_pool.size() < 2) {
        b->ctx->packed_pool.push_back(npr_ctx::HostWords{std::move(b->packed), b->packed_cap});
    } else if (b->packed && !b->ctx->packed_pool.empty()) {  // the pool keeps the larger ones
        auto &smallest = *std::min_element(b->ctx->packed_pool.begin(), b->ctx->packed_pool.end(),
                                           [](const npr_ctx::HostWords &x, const npr_ctx::HostWords &y) { return x.cap < y.cap; });
        if (smallest.cap < b->packed_cap) smallest.p = std::move(b->packed), smallest.cap = b->packed_cap;
    }
    delete b;
}

int32_t npr_batch_get_stats(const npr_batch *b, npr_batch_stats *st) {
    if (!b || !st) return NPR_ERR_INVALID;
    *st = b->stats;
    return NPR_OK;
}

int32_t npr_batch_results(const npr_batch *b, npr_read_result *out) {
    if (!b || (!out && b->n_reads)) return NPR_ERR_INVALID;
    if (!b->finished) return NPR_ERR_STATE;
    std::copy(b->results.begin(), b->results.end(), out);
    return NPR_OK;
}

static void ops_from_guide(npr_batch *b) {  // rescore mode: the guide's operations of non-zero length, in the pairs form
    const int64_t total = b->ops_off[b->n_reads];
    if (2 * total > b->ops_cap) b->ops.reset(new int32_t[2 * total]), b->ops_cap = 2 * total;
    parallel_for(b->n_reads, b->ctx->host_threads, [&](int64_t i) {
        if (b->results[i].status != NPR_OK) return;
        const int32_t *g = b->guide_ops.data() + 2 * b->guide_off[i];
        const int64_t ng = b->guide_off[i + 1] - b->guide_off[i];
        int32_t *out = b->ops.get() + 2 * b->ops_off[i];
        for (int64_t q = 0; q < ng; ++q)
            if (g[2 * q + 1] > 0) *out++ = g[2 * q], *out++ = g[2 * q + 1];
    });
    b->have_pairs_form = true;
}
static void ensure_pairs_form(npr_batch *b) {
    if (b->have_pairs_form) return;
    if (b->ops_from_guide) return ops_from_guide(b);
    const int64_t total = b->ops_off[b->n_reads];
    if (2 * total > b->ops_cap) b->ops.reset(new int32_t[2 * total]), b->ops_cap = 2 * total;
    const uint32_t *src = b->packed.get();
    int32_t *out = b->ops.get();
    const int64_t chunk = 1 << 19, nchunks = (total + chunk - 1) / chunk;
    parallel_for(nchunks, b->ctx->host_threads, [&](int64_t c) {
        for (int64_t i = c * chunk, hi = std::min(total, (c + 1) * chunk); i < hi; ++i)
            out[2 * i] = static_cast<int32_t>(src[i] & 3u), out[2 * i + 1] = static_cast<int32_t>(src[i] >> 2);
    });
    b->have_pairs_form = true;
}
static void ensure_packed_form(npr_batch *b) {
    if (b->have_packed_form) return;
    if (b->ops_from_guide && !b->have_pairs_form) ops_from_guide(b);
    const int64_t total = b->ops_off[b->n_reads];
    if (total > b->packed_cap) b->packed.reset(new uint32_t[total]), b->packed_cap = total;
    const int32_t *src = b->ops.get();
    uint32_t *out = b->packed.get();
    const int64_t chunk = 1 << 19, nchunks = (total + chunk - 1) / chunk;
    parallel_for(nchunks, b->ctx->host_threads, [&](int64_t c) {
        for (int64_t i = c * chunk, hi = std::min(total, (c + 1) * chunk); i < hi; ++i)
            out[i] = static_cast<uint32_t>(src[2 * i + 1]) << 2 | static_cast<uint32_t>(src[2 * i]);
    });
    b->have_packed_form = true;
}

int32_t npr_batch_ops(const npr_batch *b, int64_t *ops_off, int32_t *ops, int64_t cap_pairs) {
    if (!b || !ops_off) return NPR_ERR_INVALID;
    if (!b->finished) return NPR_ERR_STATE;
    std::copy(b->ops_off.begin(), b->ops_off.end(), ops_off);
    if (!ops) return NPR_OK;
    if (cap_pairs < b->ops_off[b->n_reads]) return NPR_ERR_CAPACITY;
    try {
        ensure_pairs_form(const_cast<npr_batch *>(b));
    } catch (const std::exception &) {
        return fail(b->ctx, NPR_ERR_NOMEM, "npr_batch_ops: out of host memory");
    }
    std::copy(b->ops.get(), b->ops.get() + b->ops_words, ops);
    return NPR_OK;
}

int32_t npr_batch_ops_packed(const npr_batch *b, int64_t *ops_off, uint32_t *words, int64_t cap_words) {
    if (!b || !ops_off) return NPR_ERR_INVALID;
    if (!b->finished) return NPR_ERR_STATE;
    std::copy(b->ops_off.begin(), b->ops_off.end(), ops_off);
    if (!words) return NPR_OK;
    const int64_t total = b->ops_off[b->n_reads];
    if (cap_words < total) return NPR_ERR_CAPACITY;
    try {
        ensure_packed_form(const_cast<npr_batch *>(b));
    } catch (const std::exception &) {
        return fail(b->ctx, NPR_ERR_NOMEM, "npr_batch_ops_packed: out of host memory");
    }
    // (150 MB for a chunk of 12 500 reads, into pages the caller has not touched yet: one thread took 30 ms of the job's tail)
    const uint32_t *src = b->packed.get();
    const int64_t chunk = 1 << 20, nchunks = (total + chunk - 1) / chunk;
    parallel_for(nchunks, b->ctx->host_threads, [&](int64_t c) {
        std::memcpy(words + c * chunk, src + c * chunk, sizeof(uint32_t) * static_cast<size_t>(std::min(total, (c + 1) * chunk) - c * chunk));
    });
    return NPR_OK;
}

int32_t npr_batch_pairs(const npr_batch *b, int64_t *pair_off, int32_t *x, int32_t *y, float *p, int64_t cap) {
    if (!b || !pair_off) return NPR_ERR_INVALID;
    if (!b->finished) return NPR_ERR_STATE;
    std::copy(b->pair_off.begin(), b->pair_off.end(), pair_off);
    if (!x) return NPR_OK;
    if (!b->pairs_ready) {  // realign mode left them on the device
        int32_t rc;
        try {
            rc = fetch_pairs(const_cast<npr_batch *>(b));
        } catch (const std::exception &) {
            rc = fail(b->ctx, NPR_ERR_NOMEM, "npr_batch_pairs: out of host memory");
        }
        if (rc != NPR_OK) return rc;
    }
    const int64_t total = b->pair_off[b->n_reads];
    if (cap < total) return NPR_ERR_CAPACITY;
    for (int64_t r = 0; r < b->n_reads; ++r) {  // internal coordinates are relative to the guide's window
        const int32_t gx = static_cast<int32_t>(b->gstart[2 * r]), gy = static_cast<int32_t>(b->gstart[2 * r + 1]);
        for (int64_t i = b->pair_off[r]; i < b->pair_off[r + 1]; ++i) x[i] = b->pairs[i].x + gx, y[i] = b->pairs[i].y + gy, p[i] = b->pairs[i].p;
    }
    return NPR_OK;
}

int32_t npr_batch_debug_set_pairs(npr_batch *b, int64_t read, const int32_t *x, const int32_t *y, const float *p, int64_t n, int32_t task_status) {
    if (!b || read < 0 || read >= b->n_reads || n < 0 || (n > 0 && (!x || !y || !p))) return NPR_ERR_INVALID;
    npr_ctx *ctx = b->ctx;
    if (!b->ran) return fail(ctx, NPR_ERR_STATE, "npr_batch_debug_set_pairs before npr_batch_run");
    if (b->read_ntasks[read] != 1) return fail(ctx, NPR_ERR_INVALID, "npr_batch_debug_set_pairs: the read has more than one segment");
    const int32_t k = b->task_of[b->read_first_task[read]];
    const Task &tk = b->tasks[k];
    if (n > tk.pair_cap) return NPR_ERR_CAPACITY;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n) {
        HIP_TRY(ctx, hipMemcpy(b->d_px.p + tk.pair_off, x, sizeof(int32_t) * n, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(b->d_py.p + tk.pair_off, y, sizeof(int32_t) * n, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(b->d_pp.p + tk.pair_off, p, sizeof(float) * n, hipMemcpyHostToDevice));
    }
    TaskOut o;
    HIP_TRY(ctx, hipMemcpy(&o, b->d_outs.p + k, sizeof(TaskOut), hipMemcpyDeviceToHost));
    o.npairs = static_cast<int32_t>(n), o.status = task_status;
    HIP_TRY(ctx, hipMemcpy(b->d_outs.p + k, &o, sizeof(TaskOut), hipMemcpyHostToDevice));
    b->finished = false;
    return NPR_OK;
}

int32_t npr_batch_expectations(npr_batch *b, double *T_exp, double *E_exp, double *loglik, float *kernel_ms) {
    if (!b || !T_exp || !E_exp || !loglik) return NPR_ERR_INVALID;
    npr_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> arena_lock(ctx->arena->mu);  // the E-step keeps its forward rows in the arena
    ++ctx->arena->epoch;
    std::fill(T_exp, T_exp + NPR_MAX_MODELS * 25, 0.0);
    std::fill(E_exp, E_exp + NPR_MAX_MODELS * 80, 0.0);
    std::fill(loglik, loglik + NPR_MAX_MODELS, 0.0);
    if (kernel_ms) *kernel_ms = 0.f;
    const int64_t ntasks = static_cast<int64_t>(b->tasks.size());
    if (!ntasks) return NPR_OK;
    if (b->variable_regions)
        return fail(ctx, NPR_ERR_STATE, "npr_batch_expectations: this batch was laid out for realignment only (scratch regions of their own size); "
                                        "stage it with NPR_MODE_EXPECTATIONS");
    {
        const int32_t rc = ensure_coff(b);  // classes without a register E-step take the generic kernel
        if (rc != NPR_OK) return rc;
    }
    // launch geometry: everything goes through the generic kernel (LDS ring while the band fits, global ring beyond)
    struct L {
        int first, count, wcap, grid;
        size_t lds;
        bool global_ring;
        int stair_R;  // > 0: the register-kernel E-step (k_em_stair<R>), else the generic kernel
        int wide_NW;  // > 0: stair_R slots per lane on wide_NW wavefronts per task (k_dp_wide<R, NW, EM>)
        bool tile;    // the stripe-kernel E-step (k_em_tile<stair_R>): scratch regions per workgroup, as in the DP launch
        int slot_base;    // first uniform forward-scratch region: the one its class had in the DP launch (the classes run concurrently)
        int dp_grid;      // ... and how many of them that launch owned
        int64_t cells;
        int region_first;  // stripe class: its scratch regions in the batch's table
        size_t fx_off, ring_off;  // where its planes of the other four states / its HBM ring start (floats)
    };
    std::vector<L> launches;
    int64_t max_grid = 1;
    for (const auto &dl : b->launches) {  // one E-step launch per kernel class of the batch (tasks are grouped by class)
        L l{};
        l.first = dl.first, l.count = dl.count;
        l.slot_base = dl.slot_base, l.dp_grid = dl.grid, l.cells = dl.cells, l.region_first = dl.own_regions ? dl.region_first : -1;
        if (is_one_wave_kind(kClassTab[dl.cls].kind) && ctx->opt[NPR_OPT_EM_GENERIC] == 0) {
            // 127 / 161 / 223 VGPRs and 9 KiB of LDS bins per wavefront: 16 / 12 / 8 wavefronts per CU
            l.stair_R = kClassTab[dl.cls].R;
            l.lds = em_stair_lds_bytes();
            int em_waves = l.stair_R == 4 ? 8 : (l.stair_R == 2 ? 12 : 16);
            if (ctx->opt[NPR_OPT_EM_WAVES] > 0) em_waves = static_cast<int>(std::min<int64_t>(32, ctx->opt[NPR_OPT_EM_WAVES]));  // bring-up
            l.grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(l.count, static_cast<int64_t>(ctx->cu_count) * em_waves)));
            launches.push_back(l);
            continue;
        }
        if (is_tile_kind(kClassTab[dl.cls].kind) && kClassTab[dl.cls].R == 2 && ctx->opt[NPR_OPT_EM_GENERIC] == 0) {
            // 164 VGPRs: 3 wavefronts per SIMD, 12 per CU -> 3 workgroups of 4; the workgroups keep the scratch regions the DP
            // launch gave them (region i is sized for task i, and everything the queue hands out later is smaller)
            l.stair_R = 2, l.tile = true;
            l.lds = em_tile_lds_bytes(em_tile_waves());
            l.grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(l.count, dl.grid), static_cast<int64_t>(ctx->cu_count) * (em_tile_waves_per_cu() / em_tile_waves()))));
            launches.push_back(l);
            continue;
        }
        if (kClassTab[dl.cls].kind == K_WIDE && kClassTab[dl.cls].R == 2 && ctx->opt[NPR_OPT_EM_GENERIC] == 0) {
            // 157 VGPRs: 3 wavefronts per SIMD, 12 per CU -> 3 / 1 tasks per CU on 4 / 8 wavefronts each
            l.stair_R = 2, l.wide_NW = kClassTab[dl.cls].NW;
            const int per_cu = 12 / l.wide_NW;
            l.lds = em_wide_lds_bytes(l.wide_NW);
            l.grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(l.count, static_cast<int64_t>(ctx->cu_count) * per_cu)));
            launches.push_back(l);
            continue;
        }
        l.wcap = static_cast<int>((std::max<int64_t>(dl.width, 64) + 3) & ~int64_t(3));
        l.lds = generic_lds_bytes(l.wcap) + em_extra_lds_bytes();
        l.global_ring = l.lds > 160 * 1024;  // the bins take 12 KiB of the LDS the ring would otherwise have
        if (l.global_ring) l.lds = generic_lds_bytes(0) + em_extra_lds_bytes();
        const int waves = l.global_ring ? 8 : std::min<int>(12, static_cast<int>(std::max<size_t>(1, (160 * 1024) / (l.lds + 256))));
        l.grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(l.count, static_cast<int64_t>(ctx->cu_count) * waves)));
        launches.push_back(l);
    }
    // The launches run concurrently, like the DP launches of npr_batch_run (serialised, a batch in the trainer's band spent
    // 63 ms where its longest class takes 38: profiles/r03_em_*): each class keeps the forward-scratch regions its DP launch
    // owned (so at most that many workgroups) and gets its own planes and ring.
    for (auto &l : launches)
        if (!l.tile) l.grid = std::max(1, std::min(l.grid, l.dp_grid));
    (void)max_grid;
    // The planes of the other four states: 16 bytes per cell of forward scratch in use.  The stripe kernel's mirror its regions
    // of the forward scratch, but only those of the workgroups the E-step launches (far fewer than the DP launch had): when
    // the device has no room for them, fewer workgroups yet.
    hipError_t e;
    size_t ring_floats = 0;
    for (;;) {
        // uniform classes: planes packed one class after the other; the stripe class: a mirror of its scratch regions, which
        // lie behind all uniform regions of the arena (so behind the packed planes too)
        size_t fx_cells = 0;
        ring_floats = 0;
        for (auto &l : launches) {
            if (l.tile) continue;
            l.fx_off = fx_cells;
            fx_cells += static_cast<size_t>(l.grid) * 4 * static_cast<size_t>(b->slot_stride);
            l.ring_off = ring_floats;
            if (l.global_ring) ring_floats += static_cast<size_t>(l.grid) * 18 * l.wcap;
        }
        for (auto &l : launches)
            if (l.tile && !b->region_end.empty()) {
                l.fx_off = 0;
                fx_cells = std::max(fx_cells, 4 * static_cast<size_t>(b->region_end[std::min<size_t>(static_cast<size_t>(l.grid), b->region_end.size()) - 1]));
            }
        if (fx_cells <= ctx->arena_fx_cells) break;
        if (ctx->arena_Fx) (void)hipFree(reinterpret_cast<char *>(ctx->arena_Fx) - npr_ctx::kArenaPad);
        ctx->arena_Fx = nullptr, ctx->arena_fx_cells = 0;
        char *raw = nullptr;
        e = hipMalloc(reinterpret_cast<void **>(&raw), fx_cells * sizeof(float) + 2 * npr_ctx::kArenaPad);
        if (e != hipSuccess && !ctx->cache.empty()) {  // the buffers kept from closed batches are in the way
            (void)hipGetLastError();
            ctx->cache_flush();
            e = hipMalloc(reinterpret_cast<void **>(&raw), fx_cells * sizeof(float) + 2 * npr_ctx::kArenaPad);
        }
        if (e == hipSuccess) {
            ctx->arena_Fx = reinterpret_cast<float *>(raw + npr_ctx::kArenaPad);
            ctx->arena_fx_cells = fx_cells;
            break;
        }
        (void)hipGetLastError();
        bool shrunk = false;
        for (auto &l : launches)
            if (l.grid > 1) l.grid = (l.grid + 1) / 2, shrunk = true;
        if (!shrunk) return fail(ctx, NPR_ERR_NOMEM, "npr_batch_expectations: hipMalloc of the forward planes", e);
    }
    DevBuf<float> ring;
    DevBuf<double> d_T, d_E;
    if ((e = ring.alloc(ring_floats)) != hipSuccess || (e = d_T.alloc(NPR_MAX_MODELS * 25)) != hipSuccess ||
        (e = d_E.alloc(NPR_MAX_MODELS * EM_BINS)) != hipSuccess)
        return fail(ctx, NPR_ERR_NOMEM, "npr_batch_expectations: hipMalloc", e);
    HIP_TRY(ctx, hipMemsetAsync(d_T.p, 0, d_T.bytes(), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(d_E.p, 0, d_E.bytes(), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(b->d_queue.p, 0, sizeof(int32_t) * kQueueSlots, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    // all classes at once, the smallest first, each on its own stream; the main stream waits for all of them, so
    // ev0 -> ev1 brackets the whole E-step
    std::vector<const L *> order;
    for (const auto &l : launches) order.push_back(&l);
    std::stable_sort(order.begin(), order.end(), [](const L *x, const L *y) { return x->cells < y->cells; });
    const bool serial = ctx->opt[NPR_OPT_EM_SERIAL] != 0;  // A/B switch: one launch after the other, as before round 3
    for (size_t i = 0; i < order.size(); ++i) {
        const L &l = *order[i];
        const bool last = serial || i + 1 == order.size();
        hipStream_t st = last ? ctx->stream : ctx->side[i % npr_ctx::kSideStreams];
        if (!last) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev0, 0));
        KernelArgs a = make_args(b);
        a.tasks += l.first;
        a.outs += l.first;
        a.ntasks = l.count;
        a.queue += static_cast<int>(i);  // at most kClasses launches, kQueueSlots counters
        a.wcap = l.wcap;
        a.slot_base = l.slot_base;
        a.region = l.region_first >= 0 ? b->d_region.p + l.region_first : nullptr;
        a.ring = ring.p ? ring.p + l.ring_off : nullptr;
        // stair / wide / generic kernels index their planes by workgroup from a.Fx; the stripe kernel by its scratch region
        a.Fx = ctx->arena_Fx + l.fx_off;
        a.em_T = d_T.p;
        a.em_E = d_E.p;
        const int rc = l.tile      ? launch_em_tile(a, l.stair_R, l.grid, st)
                       : l.wide_NW ? launch_em_wide(a, l.stair_R, l.wide_NW, l.grid, st)
                       : l.stair_R ? launch_em_stair(a, l.stair_R, l.grid, st)
                                   : launch_em(a, l.grid, l.lds, l.global_ring, st);
        if (rc != 0) return fail(ctx, NPR_ERR_HIP, "E-step kernel launch", static_cast<hipError_t>(rc));
        if (!last) HIP_TRY(ctx, hipEventRecord(ctx->side_done[i % npr_ctx::kSideStreams], st));
    }
    if (!serial)
        for (size_t i = 0; i + 1 < order.size(); ++i) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_done[i % npr_ctx::kSideStreams], 0));
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (kernel_ms) HIP_TRY(ctx, hipEventElapsedTime(kernel_ms, ctx->ev0, ctx->ev1));
    std::vector<double> hE(NPR_MAX_MODELS * EM_BINS);
    HIP_TRY(ctx, hipMemcpy(T_exp, d_T.p, d_T.bytes(), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(hE.data(), d_E.p, d_E.bytes(), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(b->outs.data(), b->d_outs.p, b->d_outs.bytes(), hipMemcpyDeviceToHost));
    for (int m = 0; m < NPR_MAX_MODELS; ++m) {
        const double *s = hE.data() + m * EM_BINS;
        double *d = E_exp + m * 80;
        for (int i = 0; i < 16; ++i) d[i] = s[i];
        for (int x = 0; x < 4; ++x)
            for (int y = 0; y < 4; ++y) {
                d[16 + x * 4 + y] = 0.25 * s[16 + x];  // shortGapX: count of reference base x
                d[48 + x * 4 + y] = 0.25 * s[20 + x];  // longGapX
                d[32 + x * 4 + y] = 0.25 * s[24 + y];  // shortGapY: count of read base y
                d[64 + x * 4 + y] = 0.25 * s[28 + y];  // longGapY
            }
    }
    const double LN2 = 0.69314718055994530942;
    for (int64_t k = 0; k < ntasks; ++k) {
        const TaskOut &o = b->outs[k];
        if (o.status != NPR_OK) return fail(ctx, o.status, "npr_batch_expectations: a segment has zero probability under the model");
        loglik[b->tasks[k].model] += (std::log2(static_cast<double>(o.tot_m)) + o.tot_e) * LN2;
    }
    b->ran = false;  // the task outputs now belong to the E-step
    return NPR_OK;
}

namespace {

// the kernel over n reads whose cigars are either packed on the device already (d_ops / d_off) or given on the host
int32_t run_align_stats(npr_ctx *ctx, int64_t n, const uint32_t *d_ops, const int64_t *d_off, const std::vector<uint32_t> *h_ops,
                        const std::vector<int64_t> *h_off, const std::vector<int32_t> &seg_off, const std::vector<StatsSeg> &segs,
                        const uint8_t *d_seq, int32_t *stats) {
    if (n >= (int64_t(1) << 31)) return fail(ctx, NPR_ERR_INVALID, "npr_align_stats: too many reads");
    DevBuf<uint32_t> ops;
    DevBuf<int64_t> off;
    DevBuf<int32_t> so, out;
    DevBuf<StatsSeg> sg;
    hipError_t e;
    if (!d_ops) {
        if ((e = ops.alloc(h_ops->size())) != hipSuccess || (e = off.alloc(h_off->size())) != hipSuccess)
            return fail(ctx, NPR_ERR_NOMEM, "npr_align_stats: hipMalloc", e);
        if (!h_ops->empty()) HIP_TRY(ctx, hipMemcpyAsync(ops.p, h_ops->data(), ops.bytes(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(off.p, h_off->data(), off.bytes(), hipMemcpyHostToDevice, ctx->stream));
        d_ops = ops.p, d_off = off.p;
    }
    if ((e = so.alloc(seg_off.size())) != hipSuccess || (e = sg.alloc(segs.size())) != hipSuccess ||
        (e = out.alloc(static_cast<size_t>(n) * NPR_STATS_WORDS)) != hipSuccess)
        return fail(ctx, NPR_ERR_NOMEM, "npr_align_stats: hipMalloc", e);
    HIP_TRY(ctx, hipMemcpyAsync(so.p, seg_off.data(), so.bytes(), hipMemcpyHostToDevice, ctx->stream));
    if (!segs.empty()) HIP_TRY(ctx, hipMemcpyAsync(sg.p, segs.data(), sg.bytes(), hipMemcpyHostToDevice, ctx->stream));
    StatsArgs a{static_cast<int32_t>(n), d_off, d_ops, so.p, sg.p, d_seq, out.p};
    const int rc = launch_align_stats(a, ctx->stream);
    if (rc != 0) return fail(ctx, NPR_ERR_HIP, "k_align_stats launch", static_cast<hipError_t>(rc));
    HIP_TRY(ctx, hipMemcpyAsync(stats, out.p, out.bytes(), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return NPR_OK;
}

}  // namespace

int32_t npr_batch_align_stats(npr_batch *b, int32_t *stats) {
    if (!b || (!stats && b->n_reads)) return NPR_ERR_INVALID;
    if (!b->finished) return NPR_ERR_STATE;
    npr_ctx *ctx = b->ctx;
    try {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        const int64_t n = b->n_reads;
        if (n == 0) return NPR_OK;
        // the pieces of every read's window whose base codes the batch holds: its tasks' segments
        std::vector<int32_t> seg_off(n + 1, 0);
        for (int64_t i = 0; i < n; ++i) seg_off[i + 1] = seg_off[i] + b->read_ntasks[i];
        std::vector<StatsSeg> segs(seg_off[n]);
        for (int64_t i = 0; i < n; ++i)
            for (int32_t s = 0; s < b->read_ntasks[i]; ++s) {
                const Task &t = b->tasks[b->task_of[b->read_first_task[i] + s]];
                segs[seg_off[i] + s] = StatsSeg{t.xs, t.xs + t.lX, t.ys, t.ys + t.lY, t.x_off, t.y_off};
            }
        int32_t rc;
        std::unique_lock<std::mutex> arena_lock(ctx->arena->mu);  // the resident cigars lie in the arena
        if (b->dev_ops && b->dev_ops_epoch == ctx->arena->epoch) {
            rc = run_align_stats(ctx, n, b->dev_ops, b->dev_od, nullptr, nullptr, seg_off, segs, b->d_seq.p, stats);
            arena_lock.unlock();
        } else {
            arena_lock.unlock();
            ensure_packed_form(b);
            std::vector<uint32_t> packed(b->packed.get(), b->packed.get() + b->ops_off[n]);
            rc = run_align_stats(ctx, n, nullptr, nullptr, &packed, &b->ops_off, seg_off, segs, b->d_seq.p, stats);
        }
        if (rc != NPR_OK) return rc;
        for (int64_t i = 0; i < n; ++i)
            if (b->results[i].status != NPR_OK) std::fill(stats + i * NPR_STATS_WORDS, stats + (i + 1) * NPR_STATS_WORDS, 0), stats[i * NPR_STATS_WORDS + 14] = b->results[i].status;
        return NPR_OK;
    } catch (const std::exception &) {
        return fail(ctx, NPR_ERR_NOMEM, "npr_batch_align_stats: out of host memory");
    }
}

int32_t npr_align_stats(npr_ctx *ctx, int64_t n, int64_t n_refs, const uint8_t *ref, const int64_t *ref_off, const int32_t *ref_index,
                        const uint8_t *read, const int64_t *read_off, const int32_t *ops, const int64_t *ops_off, const int64_t *start,
                        int32_t *stats) {
    if (!ctx || n < 0 || n_refs < 0 || (n && (!ref_off || !read_off || !ops_off || !stats))) return NPR_ERR_INVALID;
    if (!ref_index && n_refs != n) return fail(ctx, NPR_ERR_INVALID, "npr_align_stats: without ref_index, n_refs must equal n_reads");
    if (n == 0) return NPR_OK;
    try {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        // every read's window (the reference / read bases its cigar consumes) encoded into one code buffer
        std::vector<int64_t> woff(n + 1, 0), off(ops_off, ops_off + n + 1);
        std::vector<int32_t> seg_off(n + 1), bad(n, 0);
        std::vector<StatsSeg> segs(n);
        std::vector<int64_t> cx(n), cy(n);
        parallel_for(n, ctx->host_threads, [&](int64_t i) {
            int64_t x = 0, y = 0;
            for (int64_t q = ops_off[i]; q < ops_off[i + 1]; ++q) {
                const int32_t op = ops[2 * q], len = ops[2 * q + 1];
                if (op < 0 || op > 2 || len < 0) bad[i] = 1;
                if (op != NPR_OP_I) x += len;
                if (op != NPR_OP_D) y += len;
            }
            const int64_t k = ref_index ? ref_index[i] : i;
            const int64_t sx = start ? start[2 * i] : 0, sy = start ? start[2 * i + 1] : 0;
            if (k < 0 || k >= n_refs || sx < 0 || sy < 0 || sx + x > ref_off[k + 1] - ref_off[k] || sy + y > read_off[i + 1] - read_off[i] ||
                x >= (int64_t(1) << 30) || y >= (int64_t(1) << 30))
                bad[i] = 1;
            cx[i] = bad[i] ? 0 : x, cy[i] = bad[i] ? 0 : y;
        });
        for (int64_t i = 0; i < n; ++i) woff[i + 1] = woff[i] + cx[i] + cy[i], seg_off[i] = static_cast<int32_t>(i);
        seg_off[n] = static_cast<int32_t>(n);
        const std::unique_ptr<uint8_t[]> codes(new uint8_t[woff[n] + 1]);
        std::vector<uint32_t> packed(ops_off[n]);
        parallel_for(n, ctx->host_threads, [&](int64_t i) {
            const int64_t k = ref_index ? ref_index[i] : i;
            const int64_t sx = start ? start[2 * i] : 0, sy = start ? start[2 * i + 1] : 0;
            uint8_t *w = codes.get() + woff[i];
            if (!bad[i]) {
                const uint8_t *xs = ref + ref_off[k] + sx, *ys = read + read_off[i] + sy;
                for (int64_t q = 0; q < cx[i]; ++q) w[q] = encode_base(xs[q]);
                for (int64_t q = 0; q < cy[i]; ++q) w[cx[i] + q] = encode_base(ys[q]);
            }
            segs[i] = StatsSeg{0, static_cast<int32_t>(cx[i]), 0, static_cast<int32_t>(cy[i]), woff[i], woff[i] + cx[i]};
            for (int64_t q = ops_off[i]; q < ops_off[i + 1]; ++q)
                packed[q] = bad[i] ? 0u : (static_cast<uint32_t>(ops[2 * q + 1]) << 2 | static_cast<uint32_t>(ops[2 * q]));
        });
        DevBuf<uint8_t> d_codes;
        if (d_codes.alloc(woff[n] + 1) != hipSuccess) return fail(ctx, NPR_ERR_NOMEM, "npr_align_stats: hipMalloc");
        HIP_TRY(ctx, hipMemcpyAsync(d_codes.p, codes.get(), woff[n] + 1, hipMemcpyHostToDevice, ctx->stream));
        const int32_t rc = run_align_stats(ctx, n, nullptr, nullptr, &packed, &off, seg_off, segs, d_codes.p, stats);
        if (rc != NPR_OK) return rc;
        for (int64_t i = 0; i < n; ++i)
            if (bad[i]) std::fill(stats + i * NPR_STATS_WORDS, stats + (i + 1) * NPR_STATS_WORDS, 0), stats[i * NPR_STATS_WORDS + 14] = NPR_ERR_INVALID;
        return NPR_OK;
    } catch (const std::exception &) {
        return fail(ctx, NPR_ERR_NOMEM, "npr_align_stats: out of host memory");
    }
}

int64_t npr_batch_plan_check(npr_batch *b, const int32_t *guide_ops) {
    if (!b || (b->n_reads && b->guide_off[b->n_reads] && !guide_ops)) return NPR_ERR_INVALID;
    npr_ctx *ctx = b->ctx;
    try {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        {
            const int32_t rc = ensure_coff(b);
            if (rc != NPR_OK) return rc;
        }
        const int64_t n = b->n_reads;
        int64_t mismatches = 0;
        std::vector<int32_t> lo, nn;
        std::vector<uint32_t> co, ctl, want_ctl;
        std::vector<Stripe> st, want_st;
        for (int64_t i = 0; i < n; ++i) {
            if (b->read_status[i] != NPR_OK && b->read_ntasks[i] == 0) continue;
            Plan plan;
            const int32_t rc = build_plan(b->params, b->ref_len[i], b->read_len[i], guide_ops + 2 * b->guide_off[i],
                                          b->guide_off[i + 1] - b->guide_off[i], plan);
            if (rc != NPR_OK || static_cast<int32_t>(plan.segs.size()) != b->read_ntasks[i]) {
                ++mismatches;
                continue;
            }
            for (int32_t s = 0; s < b->read_ntasks[i]; ++s) {
                const int32_t k = b->task_of[b->read_first_task[i] + s];
                const Task &t = b->tasks[k];
                const Segment &sg = plan.segs[s];
                bool ok = t.D == sg.D() && t.xs == sg.xs && t.ys == sg.ys && t.lX == sg.xe - sg.xs && t.lY == sg.ye - sg.ys &&
                          t.flags == ((sg.ragged_start ? 1 : 0) | (sg.ragged_end ? 2 : 0)) && b->task_cells[k] == sg.cells;
                if (ok) {
                    const size_t rows = static_cast<size_t>(t.D) + 1;
                    lo.resize(rows), nn.resize(rows), co.resize(rows);
                    HIP_TRY(ctx, hipMemcpy(lo.data(), b->d_lo.p + t.band_off, rows * 4, hipMemcpyDeviceToHost));
                    HIP_TRY(ctx, hipMemcpy(nn.data(), b->d_n.p + t.band_off, rows * 4, hipMemcpyDeviceToHost));
                    HIP_TRY(ctx, hipMemcpy(co.data(), b->d_coff.p + t.band_off, rows * 4, hipMemcpyDeviceToHost));
                    uint64_t off = 0;
                    for (size_t d = 0; d < rows && ok; ++d) {
                        ok = lo[d] == sg.lo[d] && nn[d] == sg.n[d] && co[d] == static_cast<uint32_t>(off);
                        if (!ok && std::getenv("NPR_TIMING"))
                            std::fprintf(stderr, "[npr plan check] row %zu: device lo %d n %d coff %u | host lo %d n %d coff %u\n", d, lo[d], nn[d], co[d], sg.lo[d],
                                         sg.n[d], static_cast<uint32_t>(off));
                        off += (static_cast<uint64_t>(sg.n[d]) + 3) & ~uint64_t(3);
                    }
                    if (ok && t.ctl_off >= 0) {
                        int cls = -1;  // the class the task was sorted into
                        for (const auto &L : b->launches)
                            if (k >= L.first && k < L.first + L.count) cls = L.cls;
                        ctl.resize(2 * rows), want_ctl.assign(2 * rows, 0);
                        HIP_TRY(ctx, hipMemcpy(ctl.data(), b->d_ctl.p + 2 * t.ctl_off, rows * 8, hipMemcpyDeviceToHost));
                        int64_t cells = 0;
                        ok = cls >= 0 && is_register_class(cls) && build_stair_schedule(sg, kClassTab[cls].R, kClassTab[cls].NW, want_ctl.data(), &cells) &&
                             ctl == want_ctl;
                        if (!ok && std::getenv("NPR_TIMING")) {
                            size_t q = 0;
                            while (q < 2 * rows && ctl[q] == want_ctl[q]) ++q;
                            std::fprintf(stderr, "[npr plan check] class %d, control word %zu of %zu: device %08x host %08x\n", cls, q, 2 * rows,
                                         q < 2 * rows ? ctl[q] : 0u, q < 2 * rows ? want_ctl[q] : 0u);
                        }
                    }
                    if (ok && t.tile_off >= 0) {
                        const int R = kClassTab[kTileClass].R;
                        const size_t S = static_cast<size_t>(stripes_of(sg, R)) + 1;
                        st.resize(S), want_st.assign(S, Stripe{});
                        HIP_TRY(ctx, hipMemcpy(st.data(), b->d_stripes.p + t.tile_off, S * sizeof(Stripe), hipMemcpyDeviceToHost));
                        build_stripes(sg, R, want_st.data(), nullptr);
                        ok = std::memcmp(st.data(), want_st.data(), S * sizeof(Stripe)) == 0;
                        if (ok && R == 2) {  // the packed lane masks of every row
                            const size_t nrows = static_cast<size_t>(want_st[0].K);
                            std::vector<uint32_t> rm(nrows), want_rm(nrows, 0);
                            if (nrows) HIP_TRY(ctx, hipMemcpy(rm.data(), b->d_rowmask.p + t.rowmask_off, nrows * sizeof(uint32_t), hipMemcpyDeviceToHost));
                            for (size_t q = 1; q < S; ++q)
                                for (int32_t d = want_st[q].df; d <= want_st[q].dl; ++d)
                                    want_rm[want_st[q].row0 + static_cast<uint32_t>(d - want_st[q].df)] = tile_row_word(d, sg.lo[d], sg.n[d], want_st[q].X);
                            ok = rm == want_rm;
                            if (!ok && std::getenv("NPR_TIMING")) std::fprintf(stderr, "[npr plan check] row masks differ (%zu rows)\n", nrows);
                        }
                        if (!ok && std::getenv("NPR_TIMING"))
                            for (size_t q = 0; q < S; ++q)
                                if (std::memcmp(&st[q], &want_st[q], sizeof(Stripe)) != 0) {
                                    std::fprintf(stderr, "[npr plan check] stripe entry %zu of %zu: device X %d K %d df %d dl %d row0 %u | host X %d K %d df %d dl %d row0 %u\n", q, S,
                                                 st[q].X, st[q].K, st[q].df, st[q].dl, st[q].row0, want_st[q].X, want_st[q].K, want_st[q].df, want_st[q].dl, want_st[q].row0);
                                    break;
                                }
                    }
                }
                if (!ok && mismatches < 4 && std::getenv("NPR_TIMING"))
                    std::fprintf(stderr, "[npr plan check] read %lld segment %d differs (D %d, widest band row n/a, ctl %lld, stripes %lld)\n", (long long)i, s,
                                 t.D, (long long)t.ctl_off, (long long)t.tile_off);
                mismatches += ok ? 0 : 1;
            }
        }
        return mismatches;
    } catch (const std::exception &) {
        return fail(ctx, NPR_ERR_NOMEM, "npr_batch_plan_check: out of host memory");
    }
}

int32_t npr_batch_base_expectations(npr_batch *b, const uint8_t *use, int64_t n_refs, const int64_t *ref_len, double *expect, uint8_t *seen) {
    if (!b || n_refs < 0 || (n_refs && !ref_len) || !expect || !seen) return NPR_ERR_INVALID;
    if (!b->finished) return NPR_ERR_STATE;
    npr_ctx *ctx = b->ctx;
    try {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        std::vector<int64_t> base(n_refs + 1, 0);
        for (int64_t k = 0; k < n_refs; ++k) {
            if (ref_len[k] < 0) return NPR_ERR_INVALID;
            base[k + 1] = base[k] + ref_len[k];
        }
        const int64_t rows = base[n_refs], n = b->n_reads, ntasks = static_cast<int64_t>(b->tasks.size());
        std::fill(expect, expect + 4 * rows, 0.0);
        std::fill(seen, seen + rows, uint8_t(0));
        if (!ntasks || !rows) return NPR_OK;
        std::vector<int64_t> target(n, 0);
        std::vector<uint8_t> mask(n, 0);
        for (int64_t i = 0; i < n; ++i) {
            const int64_t k = b->ref_id[i];
            const bool ok = b->results[i].status == NPR_OK && (!use || use[i]) && k >= 0 && k < n_refs &&
                            b->gstart[2 * i] + b->ref_len[i] <= ref_len[k];
            if (use && use[i] && !ok && b->results[i].status == NPR_OK) return fail(ctx, NPR_ERR_INVALID, "npr_batch_base_expectations: a read's window does not fit its reference");
            mask[i] = ok ? 1 : 0;
            target[i] = ok ? base[k] + b->gstart[2 * i] : 0;
        }
        DevBuf<unsigned long long> d_e;  // fixed-point sums (npr_stats.hip): exact, hence the same from run to run
        DevBuf<uint8_t> d_seen, d_use;
        DevBuf<int64_t> d_target;
        hipError_t e;
        if ((e = d_e.alloc(4 * rows)) != hipSuccess || (e = d_seen.alloc(rows)) != hipSuccess || (e = d_use.alloc(n)) != hipSuccess ||
            (e = d_target.alloc(n)) != hipSuccess)
            return fail(ctx, NPR_ERR_NOMEM, "npr_batch_base_expectations: hipMalloc", e);
        HIP_TRY(ctx, hipMemsetAsync(d_e.p, 0, d_e.bytes(), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(d_seen.p, 0, d_seen.bytes(), ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_use.p, mask.data(), d_use.bytes(), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_target.p, target.data(), d_target.bytes(), hipMemcpyHostToDevice, ctx->stream));
        ExpectArgs a{b->d_tasks.p, b->d_outs.p, static_cast<int32_t>(ntasks), b->d_px.p, b->d_py.p, b->d_pp.p, b->d_seq.p, d_use.p, d_target.p, d_e.p, d_seen.p};
        const int rc = launch_base_expectations(a, ctx->stream);
        if (rc != 0) return fail(ctx, NPR_ERR_HIP, "k_base_expectations launch", static_cast<hipError_t>(rc));
        static_assert(sizeof(unsigned long long) == sizeof(double), "the caller's table doubles as the staging of the fixed-point sums");
        HIP_TRY(ctx, hipMemcpyAsync(expect, d_e.p, d_e.bytes(), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(seen, d_seen.p, d_seen.bytes(), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int64_t i = 0; i < 4 * rows; ++i) {
            unsigned long long fixed;
            std::memcpy(&fixed, expect + i, sizeof(fixed));
            expect[i] = static_cast<double>(fixed) / static_cast<double>(EXPECT_FIXED_ONE);
        }
        return NPR_OK;
    } catch (const std::exception &) {
        return fail(ctx, NPR_ERR_NOMEM, "npr_batch_base_expectations: out of host memory");
    }
}

int32_t npr_batch_dense(npr_batch *b, int64_t read_index, float *Fm_v, int32_t *Fm_e, float *Bm_v, int32_t *Bm_e, int64_t cap) {
    if (!b || read_index < 0 || read_index >= b->n_reads || !Fm_v || !Fm_e || !Bm_v || !Bm_e) return NPR_ERR_INVALID;
    npr_ctx *ctx = b->ctx;
    if (b->read_status[read_index] != NPR_OK) return b->read_status[read_index];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        const int32_t rc = ensure_coff(b);  // the dense dump runs the generic kernel
        if (rc != NPR_OK) return rc;
    }
    std::lock_guard<std::mutex> arena_lock(ctx->arena->mu);  // the dump runs the read in region 0 of the arena
    ++ctx->arena->epoch;
    int64_t written = 0;
    DevBuf<float> d_Bv;
    DevBuf<int32_t> d_Be;
    DevBuf<TaskOut> d_out1;
    hipError_t e;
    if ((e = d_Bv.alloc(b->slot_stride)) != hipSuccess || (e = d_Be.alloc(b->slot_stride)) != hipSuccess || (e = d_out1.alloc(1)) != hipSuccess)
        return fail(ctx, NPR_ERR_NOMEM, "npr_batch_dense: hipMalloc", e);
    // band rows are needed to strip the row padding
    for (int32_t s = 0; s < b->read_ntasks[read_index]; ++s) {
        const int32_t k = b->task_of[b->read_first_task[read_index] + s];
        const Task &t = b->tasks[k];
        KernelArgs a = make_args(b);
        a.tasks = b->d_tasks.p + k;
        a.ntasks = 1;
        a.outs = d_out1.p;
        a.Bv = d_Bv.p;
        a.Be = d_Be.p;
        // width of this task decides LDS vs global ring
        std::vector<int32_t> wn(t.D + 1);
        HIP_TRY(ctx, hipMemcpy(wn.data(), b->d_n.p + t.band_off, sizeof(int32_t) * (t.D + 1), hipMemcpyDeviceToHost));
        const int w = (*std::max_element(wn.begin(), wn.end()) + 3) & ~3;
        const bool global_ring = w > generic_max_wcap();
        DevBuf<float> ring1;
        if (global_ring) {
            if ((e = ring1.alloc(static_cast<size_t>(18) * w)) != hipSuccess) return fail(ctx, NPR_ERR_NOMEM, "npr_batch_dense: hipMalloc", e);
            a.ring = ring1.p;
        }
        a.wcap = std::max(w, 64);
        HIP_TRY(ctx, hipMemsetAsync(b->d_queue.p, 0, sizeof(int32_t) * kQueueSlots, ctx->stream));
        const int rc = launch_generic(a, 1, 256, generic_lds_bytes(global_ring ? 0 : a.wcap), true, global_ring, ctx->stream);
        if (rc != 0) return fail(ctx, NPR_ERR_HIP, "k_dp_generic<dense> launch", static_cast<hipError_t>(rc));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<int32_t> n(t.D + 1);
        std::vector<uint32_t> co(t.D + 1);
        HIP_TRY(ctx, hipMemcpy(n.data(), b->d_n.p + t.band_off, sizeof(int32_t) * (t.D + 1), hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(co.data(), b->d_coff.p + t.band_off, sizeof(uint32_t) * (t.D + 1), hipMemcpyDeviceToHost));
        std::vector<float> fv(t.cells_pad), bv(t.cells_pad);
        std::vector<int32_t> fe(t.cells_pad), be(t.cells_pad);
        // slot 0 of the generic layout: mantissa plane, then exponent plane
        HIP_TRY(ctx, hipMemcpy(fv.data(), ctx->arena->F, sizeof(float) * t.cells_pad, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(fe.data(), ctx->arena->F + sizeof(float) * b->slot_stride, sizeof(int32_t) * t.cells_pad, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(bv.data(), d_Bv.p, sizeof(float) * t.cells_pad, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(be.data(), d_Be.p, sizeof(int32_t) * t.cells_pad, hipMemcpyDeviceToHost));
        for (int32_t d = 0; d <= t.D; ++d)
            for (int32_t j = 0; j < n[d]; ++j) {
                if (written >= cap) return NPR_ERR_CAPACITY;
                Fm_v[written] = fv[co[d] + j], Fm_e[written] = fe[co[d] + j];
                Bm_v[written] = bv[co[d] + j], Bm_e[written] = be[co[d] + j];
                ++written;
            }
    }
    b->ran = false;  // the pair buffers of this read were overwritten by the debug launch
    return NPR_OK;
}

int32_t npr_batch_rs_forward(npr_batch *b, int64_t read_index, float *Fm_v, int32_t *Fm_e, int64_t cap) {
    if (!b || read_index < 0 || read_index >= b->n_reads || !Fm_v || !Fm_e) return NPR_ERR_INVALID;
    npr_ctx *ctx = b->ctx;
    if (b->read_status[read_index] != NPR_OK) return b->read_status[read_index];
    if (!b->ran) return fail(ctx, NPR_ERR_STATE, "npr_batch_rs_forward before npr_batch_run (which sizes the forward scratch)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> arena_lock(ctx->arena->mu);  // the task runs in region 0 of the arena
    ++ctx->arena->epoch;
    DevBuf<TaskOut> d_out1;
    if (d_out1.alloc(1) != hipSuccess) return fail(ctx, NPR_ERR_NOMEM, "npr_batch_rs_forward: hipMalloc");
    bool sw = false;
    for (int sl = 0; sl < NPR_MAX_MODELS; ++sl)
        if (ctx->model_set[sl] && (ctx->models[sl].T[1 * 5 + 2] != 0.f || ctx->models[sl].T[2 * 5 + 1] != 0.f)) sw = true;
    const bool flat = !sw && flat_gap_emissions(ctx);
    int64_t written = 0;
    for (int32_t s = 0; s < b->read_ntasks[read_index]; ++s) {
        const int32_t k = b->task_of[b->read_first_task[read_index] + s];
        const Task &t = b->tasks[k];
        int R = 0;
        for (const auto &L : b->launches)
            if (k >= L.first && k < L.first + L.count && (kClassTab[L.cls].kind == K_RS || kClassTab[L.cls].kind == K_MID)) R = kClassTab[L.cls].R;
        if (R == 0 || t.ctl_off < 0) return fail(ctx, NPR_ERR_STATE, "npr_batch_rs_forward: the read has a segment that k_dp_rs does not run");
        KernelArgs a = make_args(b);
        a.tasks = b->d_tasks.p + k, a.ntasks = 1, a.outs = d_out1.p, a.slot_base = 0, a.region = nullptr;
        HIP_TRY(ctx, hipMemsetAsync(b->d_queue.p, 0, sizeof(int32_t) * kQueueSlots, ctx->stream));
        const int rc = launch_rs(a, R, 1, ctx->stream, sw, flat);
        if (rc != 0) return fail(ctx, NPR_ERR_HIP, "k_dp_rs launch", static_cast<hipError_t>(rc));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const int64_t half = rs_half_cells(static_cast<int64_t>(static_cast<uint32_t>(t.cells_pad)));
        std::vector<float> fv(static_cast<size_t>(t.cells_pad));
        std::vector<int32_t> fe(static_cast<size_t>(t.D / NPR_RS_K + 1));
        std::vector<uint32_t> ctl(2 * (static_cast<size_t>(t.D) + 1));
        HIP_TRY(ctx, hipMemcpy(fv.data(), ctx->arena->F, sizeof(float) * fv.size(), hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(fe.data(), ctx->arena->F + 4 * half, sizeof(int32_t) * fe.size(), hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(ctl.data(), b->d_ctl.p + 2 * t.ctl_off, sizeof(uint32_t) * ctl.size(), hipMemcpyDeviceToHost));
        const int rshift = stair_rshift(R);
        for (int32_t d = 0; d <= t.D; ++d) {
            const uint32_t w0 = ctl[2 * d], w1 = ctl[2 * d + 1];
            int64_t first;  // scratch cell of the row's first band cell
            int32_t n;
            if (stair_packed(R, 1)) {
                const uint32_t lo0 = w1 & 127u, lo1 = (w1 >> 7) & 127u;
                n = static_cast<int32_t>(((w1 >> 14) & 127u) + ((w1 >> 21) & 127u));
                // (word 0 is where lane 0 WOULD land: below the region's start for a row whose first lanes are outside the band)
                first = static_cast<int64_t>(static_cast<int32_t>(w0 - row_bias<2>()) >> 3) + 2 * lo1 + ((lo0 + lo1) - 2 * lo1);
            } else {
                const int32_t jlo = static_cast<int32_t>(w1 & 8191u);
                n = static_cast<int32_t>((w1 >> 13) & 8191u);
                first = static_cast<int64_t>(w0) + (jlo - ((jlo >> rshift) << rshift));
            }
            for (int32_t j = 0; j < n; ++j) {
                if (written >= cap) return NPR_ERR_CAPACITY;
                if (first + j < 0 || first + j >= static_cast<int64_t>(fv.size())) return fail(ctx, NPR_ERR_STATE, "npr_batch_rs_forward: a control word points outside the task's scratch");
                Fm_v[written] = fv[static_cast<size_t>(first + j)], Fm_e[written] = fe[static_cast<size_t>(d / NPR_RS_K)];
                ++written;
            }
        }
    }
    b->ran = false;  // the pair buffers of this read were overwritten by the debug launch
    return NPR_OK;
}

int32_t npr_realign_batch(npr_ctx *ctx, const npr_params *params, int64_t n_reads, int64_t n_refs,
                          const uint8_t *ref, const int64_t *ref_off, const int32_t *ref_index,
                          const uint8_t *read, const int64_t *read_off, const int32_t *guide_ops,
                          const int64_t *guide_off, const int32_t *model_slot, npr_read_result *results,
                          int64_t *ops_off, int32_t *ops, int64_t cap_op_pairs) {
    npr_batch *b = nullptr;
    int32_t rc = npr_batch_create(ctx, params, n_reads, n_refs, ref, ref_off, ref_index, read, read_off, guide_ops, guide_off, model_slot, &b);
    if (rc == NPR_OK) rc = npr_batch_run(b, nullptr);
    if (rc == NPR_OK) rc = npr_batch_finish(b);
    if (rc == NPR_OK && results) rc = npr_batch_results(b, results);
    if (rc == NPR_OK && ops_off) rc = npr_batch_ops(b, ops_off, ops, cap_op_pairs);
    npr_batch_destroy(b);
    return rc;
}

// --------------------------------------------------------------------------------------------------
// host logic without a GPU
// --------------------------------------------------------------------------------------------------

int32_t npr_plan_create(const npr_params *params, int64_t lX, int64_t lY, const int32_t *guide_ops, int64_t n_guide_ops, npr_plan **out) {
    if (!params || !out) return NPR_ERR_INVALID;
    std::unique_ptr<npr_plan> pl(new (std::nothrow) npr_plan);
    if (!pl) return NPR_ERR_NOMEM;
    const int32_t rc = build_plan(*params, lX, lY, guide_ops, n_guide_ops, pl->plan);
    if (rc != NPR_OK) return rc;
    *out = pl.release();
    return NPR_OK;
}

void npr_plan_destroy(npr_plan *pl) { delete pl; }

int32_t npr_plan_segments(const npr_plan *pl) { return pl ? static_cast<int32_t>(pl->plan.segs.size()) : NPR_ERR_INVALID; }

int32_t npr_plan_segment_info(const npr_plan *pl, int32_t seg, int64_t *info8) {
    if (!pl || !info8 || seg < 0 || seg >= static_cast<int32_t>(pl->plan.segs.size())) return NPR_ERR_INVALID;
    const Segment &s = pl->plan.segs[seg];
    info8[0] = s.xs, info8[1] = s.ys, info8[2] = s.xe, info8[3] = s.ye;
    info8[4] = s.ragged_start, info8[5] = s.ragged_end, info8[6] = s.D(), info8[7] = s.cells;
    return NPR_OK;
}

int32_t npr_plan_segment_band(const npr_plan *pl, int32_t seg, int32_t *lo, int32_t *n) {
    if (!pl || !lo || !n || seg < 0 || seg >= static_cast<int32_t>(pl->plan.segs.size())) return NPR_ERR_INVALID;
    const Segment &s = pl->plan.segs[seg];
    std::copy(s.lo.begin(), s.lo.end(), lo);
    std::copy(s.n.begin(), s.n.end(), n);
    return NPR_OK;
}

int32_t npr_plan_frame_schedule(const npr_plan *pl, int32_t seg, int32_t slots, int32_t slots_per_lane, int32_t *jlo,
                                int32_t *rebase, uint32_t *row_off, int64_t *cells) {
    if (!pl || seg < 0 || seg >= static_cast<int32_t>(pl->plan.segs.size()) || slots_per_lane < 1 || slots < 64 * slots_per_lane ||
        slots % (64 * slots_per_lane) != 0)
        return NPR_ERR_INVALID;
    const Segment &s = pl->plan.segs[seg];
    std::vector<uint32_t> ctl(2 * (s.D() + 1));
    int64_t c = 0;
    if (!build_stair_schedule(s, slots_per_lane, slots / (64 * slots_per_lane), ctl.data(), &c)) return NPR_ERR_BAND_TOO_WIDE;
    const bool packed = stair_packed(slots_per_lane, slots / (64 * slots_per_lane));  // npr_sched.h: the words come ready to use
    for (int64_t d = 0; d <= s.D(); ++d) {
        const uint32_t w = ctl[2 * d + 1];
        if (packed) {
            const uint32_t lo0 = w & 127u, lo1 = (w >> 7) & 127u;
            if (row_off) row_off[d] = (((ctl[2 * d] - row_bias<2>()) >> 3) + 2u * lo1) & ((1u << 29) - 1u);  // the byte offset wrapped for rows before lane lo1
            if (jlo) jlo[d] = static_cast<int32_t>(lo0 + lo1);
            if (rebase) rebase[d] = static_cast<int32_t>((w >> 28) & 3u) - 1;
            continue;
        }
        if (row_off) row_off[d] = ctl[2 * d];
        if (jlo) jlo[d] = static_cast<int32_t>(w & 8191u);
        if (rebase) rebase[d] = static_cast<int32_t>((w >> 26) & 3u) - 1;
    }
    if (cells) *cells = c;
    return NPR_OK;
}

int32_t npr_plan_stripes(const npr_plan *pl, int32_t seg, int32_t slots_per_lane, int32_t *stripes5, int32_t cap, int64_t *rows) {
    if (!pl || seg < 0 || seg >= static_cast<int32_t>(pl->plan.segs.size()) || (slots_per_lane != 2 && slots_per_lane != 4)) return NPR_ERR_INVALID;
    const Segment &s = pl->plan.segs[seg];
    const int64_t S = stripes_of(s, slots_per_lane);
    if (!stripes5) return static_cast<int32_t>(S);
    if (cap < S) return NPR_ERR_CAPACITY;
    std::vector<Stripe> tab(S + 1);
    int64_t r = 0;
    build_stripes(s, slots_per_lane, tab.data(), &r);
    for (int64_t k = 0; k < S; ++k) {
        const Stripe &st = tab[1 + k];
        stripes5[5 * k] = st.X, stripes5[5 * k + 1] = st.K, stripes5[5 * k + 2] = st.df, stripes5[5 * k + 3] = st.dl;
        stripes5[5 * k + 4] = static_cast<int32_t>(st.row0);
    }
    if (rows) *rows = r;
    return static_cast<int32_t>(S);
}

int64_t npr_mea_cigar(int64_t lX, int64_t lY, const int32_t *x, const int32_t *y, const float *p, int64_t n,
                      double gap_gamma, double match_gamma, int32_t *ops, int64_t cap_pairs, double *score) {
    if (lX < 0 || lY < 0 || n < 0 || (n && (!x || !y || !p))) return NPR_ERR_INVALID;
    std::vector<Pair> pairs(n);
    for (int64_t i = 0; i < n; ++i) pairs[i] = Pair{x[i], y[i], p[i]};
    std::sort(pairs.begin(), pairs.end(), [](const Pair &a, const Pair &d) { return a.x != d.x ? a.x < d.x : a.y < d.y; });
    std::vector<int32_t> out;
    double sc = 0.0;
    const int32_t rc = mea_cigar(lX, lY, pairs.data(), n, gap_gamma, match_gamma, out, sc);
    if (rc != NPR_OK) return rc;
    if (score) *score = sc;
    const int64_t k = static_cast<int64_t>(out.size() / 2);
    if (k > cap_pairs || (k && !ops)) return NPR_ERR_CAPACITY;
    std::copy(out.begin(), out.end(), ops);
    return k;
}

int32_t npr_rescore(const int32_t *guide_ops, int64_t n_guide_ops, const int32_t *x, const int32_t *y, const float *p, int64_t n, double *score) {
    if (!score || n < 0 || n_guide_ops < 0) return NPR_ERR_INVALID;
    std::vector<Pair> pairs(n);
    for (int64_t i = 0; i < n; ++i) pairs[i] = Pair{x[i], y[i], p[i]};
    std::sort(pairs.begin(), pairs.end(), [](const Pair &a, const Pair &d) { return a.x != d.x ? a.x < d.x : a.y < d.y; });
    *score = rescore(guide_ops, n_guide_ops, pairs.data(), n);
    return NPR_OK;
}

}  // extern "C"

namespace {
// cigar text of n op lists; op q of list i is get(i, q) -> (code, length)
template <typename Count, typename Get>
int64_t format_cigars(int64_t n, Count count, Get get, int64_t *str_off, char *out, int64_t cap) {
    static const char code[3] = {'M', 'I', 'D'};
    auto digits = [](int64_t v) { int k = 1; while (v >= 10) v /= 10, ++k; return k; };
    const int threads = usable_cpus();
    std::vector<int64_t> len(n);
    std::atomic<int> bad{0};
    parallel_for((n + 255) / 256, threads, [&](int64_t c) {
        for (int64_t i = c * 256, hi = std::min(n, (c + 1) * 256); i < hi; ++i) {
            int64_t k = 0;
            for (int64_t q = 0, m = count(i); q < m; ++q) {
                const std::pair<int32_t, int64_t> o = get(i, q);
                if (o.first < 0 || o.first > 2 || o.second < 0) bad = 1;
                k += digits(o.second) + 1;
            }
            len[i] = k ? k : 1;  // an empty cigar is "*"
        }
    });
    if (bad) return NPR_ERR_INVALID;
    str_off[0] = 0;
    for (int64_t i = 0; i < n; ++i) str_off[i + 1] = str_off[i] + len[i];
    if (!out) return str_off[n];
    if (cap < str_off[n]) return NPR_ERR_CAPACITY;
    parallel_for((n + 255) / 256, threads, [&](int64_t c) {
        for (int64_t i = c * 256, hi = std::min(n, (c + 1) * 256); i < hi; ++i) {
            char *w = out + str_off[i];
            const int64_t m = count(i);
            if (m == 0) *w = '*';
            for (int64_t q = 0; q < m; ++q) {
                const std::pair<int32_t, int64_t> o = get(i, q);
                int64_t v = o.second;
                const int k = digits(v);
                for (int j = k - 1; j >= 0; --j) w[j] = static_cast<char>('0' + v % 10), v /= 10;
                w[k] = code[o.first];
                w += k + 1;
            }
        }
    });
    return str_off[n];
}
}  // namespace

extern "C" {

int64_t npr_format_cigars(int64_t n, const int64_t *ops_off, const int32_t *ops, int64_t *str_off, char *out, int64_t cap) {
    if (n < 0 || (n && (!ops_off || !str_off)) || (n && ops_off[n] > 0 && !ops)) return NPR_ERR_INVALID;
    try {
        return format_cigars(n, [&](int64_t i) { return ops_off[i + 1] - ops_off[i]; },
                             [&](int64_t i, int64_t q) { return std::pair<int32_t, int64_t>(ops[2 * (ops_off[i] + q)], ops[2 * (ops_off[i] + q) + 1]); }, str_off, out, cap);
    } catch (const std::exception &) {
        return NPR_ERR_NOMEM;
    }
}

int64_t npr_format_sam_records(int64_t n, const char *qnames, const int64_t *qname_off, const int32_t *flag, const char *rnames,
                               const int64_t *rname_off, const int32_t *ref_index, const int64_t *pos, const int32_t *mapq,
                               const int64_t *word_off, const int64_t *n_ops, const uint32_t *words, const char *seq, const int64_t *seq_off,
                               int64_t *rec_off, char *out, int64_t cap) {
    if (n < 0 || (n && (!qnames || !qname_off || !rnames || !rname_off || !ref_index || !pos || !word_off || !n_ops || !seq || !seq_off || !rec_off)))
        return NPR_ERR_INVALID;
    try {
        static const char code[3] = {'M', 'I', 'D'};
        auto digits = [](int64_t v) { int k = 1; while (v >= 10) v /= 10, ++k; return k; };
        auto put = [](char *&w, int64_t v, int k) {
            for (int j = k - 1; j >= 0; --j) w[j] = static_cast<char>('0' + v % 10), v /= 10;
            w += k;
        };
        const int threads = usable_cpus();
        std::atomic<int> bad{0};
        std::vector<int64_t> len(n);
        // fixed part of a record: ten tabs, "*", "0", "0", "*", newline
        parallel_for((n + 255) / 256, threads, [&](int64_t c) {
            for (int64_t i = c * 256, hi = std::min(n, (c + 1) * 256); i < hi; ++i) {
                int64_t k = 0;
                for (int64_t q = 0; q < n_ops[i]; ++q) {
                    const uint32_t w = words[word_off[i] + q];
                    if ((w & 3u) > 2u) bad = 1;
                    k += digits(static_cast<int64_t>(w >> 2)) + 1;
                }
                if (n_ops[i] < 0 || pos[i] < 0 || (flag && flag[i] < 0) || (mapq && mapq[i] < 0) || ref_index[i] < 0) bad = 1;
                const int64_t r = ref_index[i] < 0 ? 0 : ref_index[i];
                len[i] = (qname_off[i + 1] - qname_off[i]) + digits(flag ? flag[i] : 0) + (rname_off[r + 1] - rname_off[r]) + digits(pos[i]) +
                         digits(mapq ? mapq[i] : 255) + (k ? k : 1) + std::max<int64_t>(seq_off[i + 1] - seq_off[i], 1) + 10 + 5;
            }
        });
        if (bad) return NPR_ERR_INVALID;
        rec_off[0] = 0;
        for (int64_t i = 0; i < n; ++i) rec_off[i + 1] = rec_off[i] + len[i];
        if (!out) return rec_off[n];
        if (cap < rec_off[n]) return NPR_ERR_CAPACITY;
        parallel_for((n + 63) / 64, threads, [&](int64_t c) {
            for (int64_t i = c * 64, hi = std::min(n, (c + 1) * 64); i < hi; ++i) {
                char *w = out + rec_off[i];
                const int64_t ql = qname_off[i + 1] - qname_off[i], r = ref_index[i], rl = rname_off[r + 1] - rname_off[r],
                              sl = seq_off[i + 1] - seq_off[i];
                std::memcpy(w, qnames + qname_off[i], static_cast<size_t>(ql)), w += ql;
                *w++ = '\t';
                put(w, flag ? flag[i] : 0, digits(flag ? flag[i] : 0));
                *w++ = '\t';
                std::memcpy(w, rnames + rname_off[r], static_cast<size_t>(rl)), w += rl;
                *w++ = '\t';
                put(w, pos[i], digits(pos[i]));
                *w++ = '\t';
                put(w, mapq ? mapq[i] : 255, digits(mapq ? mapq[i] : 255));
                *w++ = '\t';
                if (n_ops[i] == 0) *w++ = '*';
                for (int64_t q = 0; q < n_ops[i]; ++q) {
                    const uint32_t cw = words[word_off[i] + q];
                    const int64_t v = static_cast<int64_t>(cw >> 2);
                    put(w, v, digits(v));
                    *w++ = code[cw & 3u];
                }
                std::memcpy(w, "\t*\t0\t0\t", 7), w += 7;
                if (sl == 0) *w++ = '*';  // an empty SEQ is "*" in SAM
                std::memcpy(w, seq + seq_off[i], static_cast<size_t>(sl)), w += sl;
                std::memcpy(w, "\t*\n", 3), w += 3;
            }
        });
        return rec_off[n];
    } catch (const std::exception &) {
        return NPR_ERR_NOMEM;
    }
}

int64_t npr_format_cigars_packed(int64_t n, const int64_t *word_off, const int64_t *n_ops, const uint32_t *words, int64_t *str_off, char *out,
                                 int64_t cap) {
    if (n < 0 || (n && (!word_off || !n_ops || !str_off || !words))) return NPR_ERR_INVALID;
    try {
        return format_cigars(n, [&](int64_t i) { return n_ops[i]; },
                             [&](int64_t i, int64_t q) {
                                 const uint32_t w = words[word_off[i] + q];
                                 return std::pair<int32_t, int64_t>(static_cast<int32_t>(w & 3u), static_cast<int64_t>(w >> 2));
                             },
                             str_off, out, cap);
    } catch (const std::exception &) {
        return NPR_ERR_NOMEM;
    }
}

void npr_encode_bases(const uint8_t *ascii, int64_t n, uint8_t *codes) {
    for (int64_t i = 0; i < n; ++i) codes[i] = encode_base(ascii[i]);
}

}  // extern "C"
