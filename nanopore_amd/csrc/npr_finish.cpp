// npr_finish.cpp -- npr_batch_finish and what reads its results: the device MEA stage, the rescore sums, the host stage, cigars and posterior pairs (utils.py:591-609; alignmentUncertainty.py:41)
// (one of the translation units of the C ABI, include/nprealign.h; what they share: npr_api_internal.h)
#include "npr_api_internal.h"

extern "C" {

}  // extern "C"
namespace npr_impl {

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

}  // namespace npr_impl
extern "C" {

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
}  // extern "C"
namespace npr_impl {
void ensure_packed_form(npr_batch *b) {
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
}  // namespace npr_impl
extern "C" {

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


}  // extern "C"