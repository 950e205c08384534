// npr_run.cpp -- npr_batch_run: the DP launches of a staged batch and the second pass of the tasks without a range certificate; the Baum-Welch E-step; the dense dumps the tests read (cactus_realign's forward / backward pass, utils.py:587)
// (one of the translation units of the C ABI, include/nprealign.h; what they share: npr_api_internal.h)
#include "npr_api_internal.h"

extern "C" {

}  // extern "C"
namespace npr_impl {
KernelArgs make_args(npr_batch *b) {
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
}  // namespace npr_impl
extern "C" {

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
    bool scaled = b->pair_rs;  // (staged for the row- / column-scaled kernels under the models of that moment)
    for (const auto &L : b->launches) scaled |= kClassTab[L.cls].kind == K_TILE_RS;
    if (scaled)
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
        bool uses_others_regions;  // a generic launch over a class without uniform regions: not beside the others
        bool tile_cs;  // ... in column-scaled arithmetic first (k_dp_tile_cs<.., EM>); what its certificate refuses goes to k_em_tile
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
            l.tile_cs = kClassTab[dl.cls].kind == K_TILE_RS && ctx->opt[NPR_OPT_TILE_RS] != 2 && ctx->opt[NPR_OPT_EM_TILE] != 1;
            const int per_cu = l.tile_cs ? em_tile_cs_waves_per_cu() / em_tile_cs_waves() : em_tile_waves_per_cu() / em_tile_waves();
            l.grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(l.count, dl.grid), static_cast<int64_t>(ctx->cu_count) * per_cu)));
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
        if (dl.own_regions) {
            // A class laid out in scratch regions of its own (the stripe class under NPR_OPT_EM_GENERIC) has no uniform regions: the generic kernel
            // counts it in regions 0, 1, .. of the arena, which belong to the other classes' launches -- so one launch after the other then, and no
            // more workgroups than the arena has room for.  (Until round 6 the launches ran side by side there: NaN counts on a batch of four classes.)
            l.uses_others_regions = true;
            l.slot_base = 0;
            const int64_t arena_cells = b->region_end.empty() ? b->slot_stride : b->region_end.back();
            l.grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(l.grid, arena_cells / std::max<int64_t>(b->slot_stride, 1))));
            l.dp_grid = l.grid;
        }
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
    // (from the context's cache of released buffers: the trainer calls this hundreds of times on one staged batch -- no hipMalloc / hipFree per call)
    if ((e = ring.alloc_from(ctx, ring_floats)) != hipSuccess || (e = d_T.alloc_from(ctx, NPR_MAX_MODELS * 25)) != hipSuccess ||
        (e = d_E.alloc_from(ctx, NPR_MAX_MODELS * EM_BINS)) != hipSuccess)
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
    bool serial = ctx->opt[NPR_OPT_EM_SERIAL] != 0;  // A/B switch: one launch after the other, as before round 3
    for (const auto &l : launches) serial |= l.uses_others_regions;
    bool sw = false;  // (as npr_batch_run: the column-scaled kernel leaves out the short-gap switch terms no loaded model has)
    for (int sl = 0; sl < NPR_MAX_MODELS; ++sl)
        if (ctx->model_set[sl] && (ctx->models[sl].T[1 * 5 + 2] != 0.f || ctx->models[sl].T[2 * 5 + 1] != 0.f)) sw = true;
    const bool flat = !sw && flat_gap_emissions(ctx);
    for (auto &l : launches)
        if (l.tile_cs)
            for (int sl = 0; sl < NPR_MAX_MODELS; ++sl)
                if (ctx->model_set[sl] && !rs_model_ok(ctx->models[sl])) l.tile_cs = false;
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
        if (l.tile_cs) a.wcap = ctx->opt[NPR_OPT_EM_TILE] == 2 ? 1 : 0;  // (the column-scaled kernel has no ring: the field carries the test switch)
        const int rc = l.tile_cs   ? launch_em_tile_cs(a, em_tile_cs_waves(), l.grid, st, sw, flat)
                       : l.tile    ? launch_em_tile(a, l.stair_R, l.grid, st)
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
    b->outs.resize(b->tasks.size());
    HIP_TRY(ctx, hipMemcpy(b->outs.data(), b->d_outs.p, b->d_outs.bytes(), hipMemcpyDeviceToHost));
    // The tasks the column-scaled kernel did not count (TASK_RERUN: its range certificate, or backward values far above a lane's scale) are counted
    // here by k_em_tile, on the scratch regions and planes the first launch had, into the same sums.
    for (size_t i = 0; i < order.size(); ++i) {
        const L &l = *order[i];
        if (!l.tile_cs) continue;
        std::vector<int32_t> again;
        for (int k = l.first; k < l.first + l.count; ++k)
            if (b->outs[k].status == TASK_RERUN) again.push_back(k);
        if (std::getenv("NPR_TIMING")) std::fprintf(stderr, "[npr] E-step: %zu of %d stripe tasks counted again with a per-cell exponent\n", again.size(), l.count);
        if (again.empty()) continue;
        std::vector<Task> sub(again.size());
        for (size_t j = 0; j < again.size(); ++j) sub[j] = b->tasks[again[j]];
        DevBuf<Task> d_sub;
        DevBuf<TaskOut> d_subout;
        if (d_sub.alloc_from(ctx, sub.size()) != hipSuccess || d_subout.alloc_from(ctx, sub.size()) != hipSuccess) return fail(ctx, NPR_ERR_NOMEM, "npr_batch_expectations: hipMalloc");
        HIP_TRY(ctx, hipMemcpy(d_sub.p, sub.data(), sizeof(Task) * sub.size(), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemsetAsync(b->d_queue.p + static_cast<int>(i), 0, sizeof(int32_t), ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        KernelArgs a = make_args(b);
        a.tasks = d_sub.p, a.outs = d_subout.p, a.ntasks = static_cast<int32_t>(sub.size());
        a.queue += static_cast<int>(i);
        a.slot_base = l.slot_base;
        a.region = l.region_first >= 0 ? b->d_region.p + l.region_first : nullptr;  // (task j of `again` is no larger than the j-th task of the class)
        a.Fx = ctx->arena_Fx + l.fx_off;
        a.em_T = d_T.p, a.em_E = d_E.p;
        const int grid = static_cast<int>(std::min<size_t>(sub.size(), static_cast<size_t>(l.grid)));
        const int rc = launch_em_tile(a, l.stair_R, grid, ctx->stream);
        if (rc != 0) return fail(ctx, NPR_ERR_HIP, "E-step kernel launch (second pass)", static_cast<hipError_t>(rc));
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (kernel_ms) {
            float ms = 0.f;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            *kernel_ms += ms;
        }
        std::vector<TaskOut> subout(sub.size());
        HIP_TRY(ctx, hipMemcpy(subout.data(), d_subout.p, sizeof(TaskOut) * sub.size(), hipMemcpyDeviceToHost));
        for (size_t j = 0; j < again.size(); ++j) b->outs[again[j]] = subout[j];
    }
    std::vector<double> hE(NPR_MAX_MODELS * EM_BINS);
    HIP_TRY(ctx, hipMemcpy(T_exp, d_T.p, d_T.bytes(), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(hE.data(), d_E.p, d_E.bytes(), hipMemcpyDeviceToHost));
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


}  // extern "C"