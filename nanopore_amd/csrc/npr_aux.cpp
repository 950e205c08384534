// npr_aux.cpp -- post-alignment statistics on the device, base expectations of the marginAlign SNP caller, the planner cross-check (coverage.py / substitutions.py / indels.py; marginAlignSnpCaller.py:150-155)
// (one of the translation units of the C ABI, include/nprealign.h; what they share: npr_api_internal.h)
#include "npr_api_internal.h"

extern "C" {

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


}  // extern "C"