// npr_api.cpp -- the C ABI of libnprealign (include/nprealign.h): context, options, model slots, the plan-inspection entry points and the small
// public helpers.  Replaces the per-read process fan-out / temp-file gather of nanopore/analyses/utils.py:557-609 by one batched call (staging:
// npr_stage.cpp, the DP pass: npr_run.cpp, the finish: npr_finish.cpp).  There is no CPU execution path for the DP: without a usable gfx950
// device npr_create fails.
#include "npr_api_internal.h"

namespace npr_impl {
DeviceArena g_arena[kMaxDevices];
}  // namespace npr_impl

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
    // The side streams carry the SMALL launches of a pass (the classes with few cells beside the one that fills the chip): at the highest priority,
    // so that when all of a pass's launches become ready together the small ones are dispatched first -- a kernel that fills every SIMD's
    // registers with persistent workgroups leaves no room for a late one until it drains (NPR_SIDE_PRIORITY=0: default priority, A/B)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const char *pe = std::getenv("NPR_SIDE_PRIORITY");
    const int side_prio = (pe && std::atoi(pe) == 0) ? prio_lo : prio_hi;
    for (int i = 0; i < npr_ctx::kSideStreams; ++i)
        if ((e = hipStreamCreateWithPriority(&ctx->side[i], hipStreamNonBlocking, side_prio)) != hipSuccess ||
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
}  // extern "C"
namespace npr_impl {
int32_t release_scratch(npr_ctx *ctx, bool caches_only) {
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
}  // namespace npr_impl
extern "C" {

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