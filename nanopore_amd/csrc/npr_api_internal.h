// npr_api_internal.h -- what the translation units of the C ABI share (round 6: npr_api.cpp was one file of 3 000 lines):
// the context and batch structures, device buffers, the kernel class table, error plumbing.  Internal to libnprealign; not installed.
//   npr_api.cpp     context, options, models, the plan-inspection entry points, the small public helpers
//   npr_stage.cpp   npr_batch_create*: plan points, packing, H2D, the device planner, kernel classes and launch geometry
//   npr_run.cpp     npr_batch_run (launch policy, second pass of tasks without a range certificate), the E-step, the dense dumps
//   npr_finish.cpp  npr_batch_finish and what reads its results: the device MEA stage, the rescore sums, the host stage, ops / pairs
//   npr_aux.cpp     post-alignment statistics, base expectations, the planner cross-check
//   npr_text.cpp    cigar and SAM record text (the transport forms a job ships)
#pragma once
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

namespace npr_impl {
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
extern DeviceArena g_arena[kMaxDevices];  // (npr_api.cpp)
}  // namespace npr_impl
using namespace npr_impl;

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

namespace npr_impl {

// NPR_POISON=<byte>: every device buffer is filled with that byte when it is handed out (and the forward scratch before
// every batch), so that a kernel reading memory nobody wrote gives the same wrong answer on every box instead of
// depending on what the previous owner of the memory left there.  Test / bring-up switch.
inline int poison_byte() {
    const char *e = std::getenv("NPR_POISON");
    return e && e[0] ? static_cast<int>(std::strtol(e, nullptr, 0)) & 0xff : -1;
}
inline void poison(void *p, size_t bytes) {
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

inline int32_t fail(npr_ctx *ctx, int32_t code, const char *what, hipError_t e = hipSuccess) {
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

}  // namespace npr_impl
using namespace npr_impl;

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
namespace npr_impl {

inline bool build_stair_schedule(const Segment &s, int R, int NW, uint32_t *ctl, int64_t *cells) {
    if (s.n.empty()) return false;
    return stair_schedule(s.lo.data(), s.n.data(), s.D(), s.max_width, R, NW, ctl, cells);
}

}  // namespace npr_impl
using namespace npr_impl;

// Kernel classes of a batch, each launched on its own: the register kernel with one wavefront per task (R slots per
// lane), the register kernel with NW wavefronts per task (k_dp_wide), the generic kernel with an LDS ring in three
// width classes, the generic kernel with its ring in HBM.
namespace npr_impl {
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
inline bool rs_model_ok(const DevModel &m) {
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
inline bool flat_gap_emissions(const npr_ctx *ctx) {
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
inline void build_stripes(const Segment &s, int R, Stripe *out, int64_t *rows_out) {
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
}  // namespace npr_impl
using namespace npr_impl;

// --------------------------------------------------------------------------------------------------
// batch
// --------------------------------------------------------------------------------------------------


// ---- functions one translation unit defines and another calls ----
namespace npr_impl {
int32_t rescore_stage(npr_batch *b);                       // npr_finish.cpp: NPR_MODE_RESCORE_ORIGINAL, the guide's M columns as a device table (called when a batch is staged)
KernelArgs make_args(npr_batch *b);                        // npr_run.cpp: the kernel arguments of a staged batch
int32_t ensure_coff(npr_batch *b);                         // npr_stage.cpp: the generic kernel's row offsets, made on demand
int32_t release_scratch(npr_ctx *ctx, bool caches_only);   // npr_api.cpp
void ensure_packed_form(npr_batch *b);                     // npr_finish.cpp: the batch's cigars as one word per operation
}  // namespace npr_impl
