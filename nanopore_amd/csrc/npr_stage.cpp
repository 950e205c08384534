// npr_stage.cpp -- npr_batch_create*: band planning, packing, H2D, the device planner, kernel classes and launch geometry of a batch (replaces the per-read fan-out of nanopore/analyses/utils.py:557-574)
// (one of the translation units of the C ABI, include/nprealign.h; what they share: npr_api_internal.h)
#include "npr_api_internal.h"

extern "C" {

int32_t npr_batch_create(npr_ctx *ctx, const npr_params *params, int64_t n_reads, int64_t n_refs,
                         const uint8_t *ref, const int64_t *ref_off, const int32_t *ref_index,
                         const uint8_t *read, const int64_t *read_off, const int32_t *guide_ops,
                         const int64_t *guide_off, const int32_t *model_slot, npr_batch **out) {
    return npr_batch_create_at(ctx, params, n_reads, n_refs, ref, ref_off, ref_index, read, read_off, guide_ops, guide_off,
                               nullptr, model_slot, out);
}

// Row offsets of the generic kernel (rows padded to 4 cells), made on the device from the band rows the first time a
// generic launch needs them: batches whose tasks all go to the register kernels never pay for them.
}  // extern "C"
namespace npr_impl {
int32_t ensure_coff(npr_batch *b) {
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
}  // namespace npr_impl
extern "C" {

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
    // E-step batches whose stripe tasks run in column-scaled arithmetic (k_dp_tile_cs's E-step instance, below): the four-slot frame class goes there
    // too -- k_em_stair<4> is one long dependent chain per task.  (Not the two-slot class: bands of 150 / 200 cells gain 19 / 9 % on the stripes, but
    // one wavefront walks a task's stripes one after the other, and the long thin tasks of that class -- 24 000 stripe rows where the frame has
    // 16 000 anti-diagonals -- become the launch's critical path: the bench's batch 51 -> 60 ms.)
    bool em_stripes_cs = b->params.mode == NPR_MODE_EXPECTATIONS && use_tile && ctx->opt[NPR_OPT_ARITH] != 1 && ctx->opt[NPR_OPT_EM_TILE] != 1 &&
                         ctx->opt[NPR_OPT_TILE_RS] != 2;
    for (int sl = 0; sl < NPR_MAX_MODELS; ++sl)
        if (ctx->model_set[sl] && !rs_model_ok(ctx->models[sl])) em_stripes_cs = false;
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
            if (em_stripes_cs && kClassTab[c].kind == K_STAIR && kClassTab[c].R == 4) continue;
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
        bool scaled = ctx->opt[NPR_OPT_ARITH] != 1 && !force_generic;
        for (int sl = 0; sl < NPR_MAX_MODELS; ++sl)
            if (ctx->model_set[sl] && !rs_model_ok(ctx->models[sl])) scaled = false;
        // (the E-step has kernels in this arithmetic for the stripe tasks only: k_dp_tile_cs's E-step instance, NPR_OPT_EM_TILE)
        const bool em = b->params.mode == NPR_MODE_EXPECTATIONS;
        const bool rs = scaled && !em;
        b->pair_rs = rs;
        if (scaled && !(em && ctx->opt[NPR_OPT_EM_TILE] == 1))
            for (int64_t k = 0; k < ntasks; ++k) {
                if (rs && cls_of[k] >= 0 && cls_of[k] < 3) cls_of[k] = static_cast<int8_t>(kFirstRs + cls_of[k]);
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


}  // extern "C"