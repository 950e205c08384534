// npr_kernel_rs.hip -- k_dp_rs<R>: the one-wavefront frame kernel (k_dp_stair's mapping: npr_kernel_stair.hip) in
// ROW-SCALED arithmetic (npr_rs.h): one exponent per anti-diagonal row of the wavefront instead of one per cell.
//
// Same recurrences -- cactus_realign's banded five-state forward / backward / posterior pass, SURVEY.md 8a rows
// a5.3-a5.5, reference call sites nanopore/analyses/utils.py:587, alignmentUncertainty.py:41,
// marginAlignSnpCaller.py:136-146 --, same frame, same frame schedule and control words (npr_sched.h), same outputs
// (TaskOut, sparse posterior triples).  What differs from k_dp_stair is the cell: five plain fp32 values, the exponent in an
// SGPR, 18 / 20 multiply-adds per cell and direction and nothing else in the recurrence, rows of 4 bytes per cell in the
// forward scratch.  The kernel for every band a wavefront's frame can hold (classes 0-2: NPR_OPT_ARITH = 1 brings the
// per-cell-exponent kernels back for A/B runs).
#include <hip/hip_runtime.h>
#include <type_traits>

#include "npr_device.h"
#include "npr_frame.h"
#include "npr_rs.h"

namespace npr {

namespace {

template <int R, bool SW, bool FLAT>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(R == 2 ? NPR_RS_WAVES2 : 1))) k_dp_rs(KernelArgs a) {
    // static LDS: the tables' addresses are compile-time constants and fold into the ds_read offsets
    __shared__ __attribute__((aligned(16))) RsTables ltab_s;
    __shared__ __attribute__((aligned(16))) float lmodel[MODEL_FLOATS];
    __shared__ int lmisc[8];
    RsTables *ltab = &ltab_s;

    const int lane = threadIdx.x;
    int64_t fcell = static_cast<int64_t>(a.slot_base + blockIdx.x) * a.slot_stride;
    if (a.region) fcell = a.region[blockIdx.x];
    char *const F = a.F + uni64(fcell) * 8;
    const int voff = 4 * R * lane;  // byte offset of this lane's cells inside a row that starts at lane 0
    int jr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) jr[r] = R * lane + r;

    int t = blockIdx.x;
    while (t < a.ntasks) {
        const Task *tp = a.tasks + t;
        const int64_t x_off = uni64(tp->x_off), y_off = uni64(tp->y_off), ctl_off = uni64(tp->ctl_off), pair_off = uni64(tp->pair_off);
        const int lX = uni(tp->lX), lY = uni(tp->lY), D = uni(tp->D), pair_cap = min(uni(tp->pair_cap), (1 << 29) - 1) /* (pair slots are 32-bit byte offsets) */, flags = uni(tp->flags),
                  model = uni(tp->model), xs = uni(tp->xs), ys = uni(tp->ys);
        const int64_t half = rs_half_cells(static_cast<int64_t>(static_cast<uint32_t>(uni(tp->cells_pad))));
        int *const fexp = reinterpret_cast<int *>(F + 4 * half);
        const __amdgpu_buffer_rsrc_t frs = rs_task_rsrc<R>(F);
        const int rs = flags & 1, re = (flags >> 1) & 1;

        __syncthreads();
        {
            const float *gm = reinterpret_cast<const float *>(a.models + model);
            for (int i = lane; i < MODEL_FLOATS; i += WAVE) lmodel[i] = gm[i];
        }
        __syncthreads();
        rs_build_tables(ltab, reinterpret_cast<const DevModel *>(lmodel), lane, WAVE);
        __syncthreads();
        StepEnv E;
        E.mdl = reinterpret_cast<const DevModel *>(lmodel);
        E.ltab = reinterpret_cast<const char *>(ltab);
        E.X = a.seq + x_off, E.Y = a.seq + y_off, E.lX = lX, E.lY = lY, E.lane = lane;
        {
            Trans tr = load_trans(E.mdl->T);
            if constexpr (R >= NPR_RS_T_SGPR_MIN_R) {
                tr.mm = unif(tr.mm), tr.sxm = unif(tr.sxm), tr.sym = unif(tr.sym), tr.lxm = unif(tr.lxm), tr.lym = unif(tr.lym);
                tr.msx = unif(tr.msx), tr.sxsx = unif(tr.sxsx), tr.sysx = unif(tr.sysx);
                tr.msy = unif(tr.msy), tr.sysy = unif(tr.sysy), tr.sxsy = unif(tr.sxsy);
                tr.mlx = unif(tr.mlx), tr.lxlx = unif(tr.lxlx), tr.mly = unif(tr.mly), tr.lyly = unif(tr.lyly);
            }
            E.tr = tr;
        }
        const DevModel *mdl = E.mdl;

        // =============================== forward ===============================
        // A holds the even anti-diagonals, B the odd ones; X-steps lead into odd anti-diagonals, Y-steps into even ones.
        RsState<R> Q;
        Q.A = zero_rdiag<R>(), Q.B = zero_rdiag<R>();
        cptr32 ctl = (cptr32)(a.ctl + 2 * ctl_off);
        const RowCtl<R> c0 = read_row_ctl<R>(ctl, 0);
        const int j0 = c0.jlo;  // slot of the lattice point (0, 0)
        Q.x0 = -j0, Q.y0 = j0;
        Q.e = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            Q.S.X.b[r] = base8<RS_XS>(E.X, lX, Q.x0 + jr[r] - 1);
            Q.S.Y.b[r] = base8(E.Y, lY, Q.y0 - jr[r] - 1);
        }
        Q.S.xcap = RS_NX, Q.S.ycap = RS_N8;
        feed8_init<+1, RS_XS>(Q.S.fx, E.X, lX, Q.x0 + 64 * R - 1, lane);  // first X-step injects X[(x0 + 1) + 64R - 2]
        feed8_init<+1>(Q.S.fy, E.Y, lY, Q.y0, lane);               // first Y-step injects Y[(y0 + 1) - 1]
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (jr[r] == j0) {
                RCell c;
                c.m = mdl->start[rs * 5 + 0], c.sx = mdl->start[rs * 5 + 1], c.sy = mdl->start[rs * 5 + 2];
                c.lx = mdl->start[rs * 5 + 3], c.ly = mdl->start[rs * 5 + 4];
                Q.A.c[r] = c;
            }
        if (lane == 0) fexp[0] = 0;
        rs_store_row<R>(frs, Q.A, c0, voff);
        int d = 1;
        CtlPair wn = ctl_scalar2(ctl, 1);  // (two words past the task's last row at most: still inside d_ctl or its padding)
        {
            // Blocks of RS_K anti-diagonals, d = 1 (mod RS_K) at the head of each: the base streams are looked after once per block
            // (feed8_ahead) and the renormalisation needs no test -- it belongs to the block's last pair.  The pairs of a block are
            // one loop body, the last pair a second one; the rows after the last full block run through the first.
            auto pair = [&](auto last) __attribute__((always_inline)) {
                const CtlPair w = wn;
                wn = ctl_scalar2(ctl, d + 2);
                {
                    const RowCtl<R> cur = row_ctl_of_words<R>(w.a0, w.a1);
                    RS_FWD_REBASE(cur.reb);
                    rs_fwd_x_step<R, false, SW, FLAT>(E, Q.B, Q.A, Q.S, Q.x0, cur.mk, cur.moved);
                    rs_store_row<R>(frs, Q.B, cur, voff);
                }
                const RowCtl<R> cur = row_ctl_of_words<R>(w.b0, w.b1);
                RS_FWD_REBASE(cur.reb);
                rs_fwd_y_step<R, false, SW, FLAT>(E, Q.A, Q.B, Q.S, Q.y0, cur.mk, cur.moved);
                if constexpr (decltype(last)::value) {
                    Q.e += rs_renorm<R>(Q.A, Q.B);
                    if (lane == 0) fexp[(d + 1) / RS_K] = Q.e;
                }
                rs_store_row<R>(frs, Q.A, cur, voff);
                d += 2;
            };
            // (two pairs per loop body: a step moves its stream's registers on by one slot -- with two slots per lane the registers
            // trade places, and a body of ONE pair paid for that with four copies per step; after two pairs they are back)
            static_assert((RS_K / 2) % 2 == 0, "a block is a whole number of double pairs");
            while (d + RS_K - 1 <= D) {
                feed8_ahead<+1, RS_XS>(Q.S.fx, E.X, lX, Q.x0 + 64 * R - 1, lane);
                feed8_ahead<+1>(Q.S.fy, E.Y, lY, Q.y0, lane);
#pragma nounroll
                for (int k = 0; k < RS_K / 4 - 1; ++k) pair(std::false_type{}), pair(std::false_type{});
                pair(std::false_type{});
                pair(std::true_type{});
            }
            feed8_ahead<+1, RS_XS>(Q.S.fx, E.X, lX, Q.x0 + 64 * R - 1, lane);
            feed8_ahead<+1>(Q.S.fy, E.Y, lY, Q.y0, lane);
#pragma nounroll
            while (d + 1 <= D) pair(std::false_type{});
        }
        if (d <= D) {  // D odd: one more X-step, into B
            const CtlPair w = wn;
            const RowCtl<R> cur = row_ctl_of_words<R>(w.a0, w.a1);
            RS_FWD_REBASE(cur.reb);
            rs_fwd_x_step<R, true, SW, FLAT>(E, Q.B, Q.A, Q.S, Q.x0, cur.mk, cur.moved);
            rs_store_row<R>(frs, Q.B, cur, voff);
        }
        // total probability at the end corner (lX, lY): slot lX - x0 of the last anti-diagonal
        {
            const int je = lX - Q.x0;
            const bool oddD = D & 1;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (jr[r] == je) {
                    const RCell c = oddD ? Q.B.c[r] : Q.A.c[r];
                    const float raw = rs_dot5(mdl->end + re * 5, c);
                    float tm = 0.f;
                    int te = E_DEAD;
                    if (raw > 0.f) {
                        int k;
                        tm = __builtin_frexpf(raw, &k);
                        te = Q.e + k;
                    }
                    reinterpret_cast<float *>(lmisc)[0] = tm;
                    lmisc[1] = te;
                }
        }
        __syncthreads();  // (waits for the wavefront's stores too: the rows and their exponents are in L2)
        __builtin_amdgcn_s_dcache_inv();
        const float tot_m = unif(reinterpret_cast<float *>(lmisc)[0]);
        const int tot_e = uni(lmisc[1]);
        __syncthreads();

        TaskOut out;
        out.tot_m = tot_m, out.tot_e = tot_e, out.btot_m = 0.f, out.btot_e = E_DEAD, out.npairs = 0;
        out.status = NPR_OK;
        const bool alive = tot_m > 0.f;
        if (!alive) out.status = NPR_ERR_ZERO_PROB;

        // =============================== backward + posteriors ===============================
        int cnt = 0;
        int smax = -(1 << 30);  // the largest eF + eB - eTot of any anti-diagonal: the range certificate (npr_device.h)
        if (alive) {
            const float inv_tot = 1.0f / tot_m;
            const PairSink sink{a.px + pair_off, a.py + pair_off, a.pp + pair_off, 0, pair_cap, xs, ys, a.threshold};  // (32-bit slots from here)
            // A holds the even anti-diagonals again, B the odd ones; fa / fb the forward rows that pair with them, loaded one
            // anti-diagonal ahead.  S now holds X[x]*8 and Y[y]*8 of every slot.  Q.e: the backward rows' exponent.
            Q.A = zero_rdiag<R>(), Q.B = zero_rdiag<R>();
            Q.e = 0;
            const bool oddD = D & 1;
            RowCtl<R> cur = read_row_ctl<R>(ctl, D);
            // `moved` of the anti-diagonals one and two above the one being computed: the row two above is the one overwritten
            uint32_t m1 = cur.moved, m2 = 0;
            cptr_i32 fexp_c = (cptr_i32)fexp;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                Q.S.X.b[r] = base8<RS_XS>(E.X, lX, Q.x0 + jr[r]);
                Q.S.Y.b[r] = base8(E.Y, lY, Q.y0 - jr[r]);
                if (Q.x0 + jr[r] == lX) {  // the end corner (it is in the band by construction)
                    RCell c;
                    c.m = mdl->end[re * 5 + 0], c.sx = mdl->end[re * 5 + 1], c.sy = mdl->end[re * 5 + 2];
                    c.lx = mdl->end[re * 5 + 3], c.ly = mdl->end[re * 5 + 4];
                    if (oddD) Q.B.c[r] = c; else Q.A.c[r] = c;
                }
            }
            Q.S.xcap = RS_NX, Q.S.ycap = RS_N8;
            // first undone X-step injects X[x0 - 1] at slot 0; first undone Y-step injects Y[y0 - 64R] on top
            feed8_init<-1, RS_XS>(Q.S.fx, E.X, lX, Q.x0 - 1, lane);
            feed8_init<-1>(Q.S.fy, E.Y, lY, Q.y0 - 64 * R, lane);
            RFRow<R> fa, fb;
#pragma unroll
            for (int r = 0; r < R; ++r) fa.v[r] = fb.v[r] = 0.f;
            RowCtl<R> nxt = cur;
            if (oddD) {
                rs_load_row<R>(frs, fb, cur, voff);
                nxt = read_row_ctl<R>(ctl, D - 1);
                rs_load_row<R>(frs, fa, nxt, voff);
                rs_emit_pairs<R>(sink, Q.B, fb, D, Q.x0, Q.y0, cur.mk, note_s(smax, fexp_c[D / RS_K] + Q.e - tot_e), inv_tot, jr, cnt);
            } else {
                rs_load_row<R>(frs, fa, cur, voff);
                if (D >= 1) {
                    nxt = read_row_ctl<R>(ctl, D - 1);
                    rs_load_row<R>(frs, fb, nxt, voff);
                }
                rs_emit_pairs<R>(sink, Q.A, fa, D, Q.x0, Q.y0, cur.mk, note_s(smax, fexp_c[D / RS_K] + Q.e - tot_e), inv_tot, jr, cnt);
            }
            // `cur` is the control word of the anti-diagonal above the one computed next: its rebase is undone first
            int d2 = D - 1;
            if (oddD) {  // peel one even anti-diagonal so that the loop below always starts on an odd one
                const int reb = cur.reb;
                cur = nxt;
                if (d2 >= 1) {
                    nxt = read_row_ctl<R>(ctl, d2 - 1);
                    rs_load_row<R>(frs, fb, nxt, voff);
                }
                RS_BWD_REBASE(reb);
                rs_bwd_x_step<R, true, SW, FLAT>(E, Q.A, Q.B, Q.S, Q.x0, cur.mk, m2);
                m2 = m1, m1 = cur.moved;
                if ((d2 & (RS_K - 1)) == 0) Q.e += rs_renorm<R>(Q.A, Q.B);
                rs_emit_pairs<R>(sink, Q.A, fa, d2, Q.x0, Q.y0, cur.mk, note_s(smax, fexp_c[d2 / RS_K] + Q.e - tot_e), inv_tot, jr, cnt);
                d2 -= 1;
            }
            {
                // Blocks that end on a renormalising row: the pairs (d2, d2 - 1) from d2 down to RS_K m + 1 with m = d2 / RS_K; the rows of a
                // block share the forward exponent fexp[m]; the last pair ends on row RS_K m, renormalises there and is a loop body of its
                // own.  (The first block is as long as it takes to get there; the control words are read down to row -3: kCtlFrontPad.)
                CtlPair wb = ctl_scalar2(ctl, d2 - 2);
                int ef = 0, sblk = 0;
                auto pair = [&](auto last) __attribute__((always_inline)) {
                    const CtlPair q = wb;
                    wb = ctl_scalar2(ctl, d2 - 4);
                    int reb = cur.reb;
                    cur = nxt;
                    nxt = row_ctl_of_words<R>(q.b0, q.b1);
                    rs_load_row<R>(frs, fa, nxt, voff);  // for the step after this one
                    RS_BWD_REBASE(reb);
                    rs_bwd_y_step<R, false, SW, FLAT>(E, Q.B, Q.A, Q.S, Q.y0, cur.mk, m2);
                    m2 = m1, m1 = cur.moved;
                    rs_emit_pairs<R>(sink, Q.B, fb, d2, Q.x0, Q.y0, cur.mk, sblk, inv_tot, jr, cnt);
                    reb = cur.reb;
                    cur = nxt;
                    if (!decltype(last)::value || d2 >= 2) {
                        nxt = row_ctl_of_words<R>(q.a0, q.a1);
                        rs_load_row<R>(frs, fb, nxt, voff);
                    }
                    RS_BWD_REBASE(reb);
                    rs_bwd_x_step<R, false, SW, FLAT>(E, Q.A, Q.B, Q.S, Q.x0, cur.mk, m2);
                    m2 = m1, m1 = cur.moved;
                    if constexpr (decltype(last)::value) {
                        Q.e += rs_renorm<R>(Q.A, Q.B);
                        sblk = note_s(smax, ef + Q.e - tot_e);
                    }
                    rs_emit_pairs<R>(sink, Q.A, fa, d2 - 1, Q.x0, Q.y0, cur.mk, sblk, inv_tot, jr, cnt);
                    d2 -= 2;
                };
                auto block_head = [&]() __attribute__((always_inline)) {
                    ef = fexp_c[d2 / RS_K];
                    sblk = note_s(smax, ef + Q.e - tot_e);
                    feed8_ahead<-1, RS_XS>(Q.S.fx, E.X, lX, Q.x0 - 1, lane);
                    feed8_ahead<-1>(Q.S.fy, E.Y, lY, Q.y0 - 64 * R, lane);
                };
                if (d2 >= 1) {  // the first block: as long as it takes to get to a renormalising row, an odd number of pairs maybe
                    block_head();
                    const int n = (d2 & (RS_K - 1)) >> 1;
#pragma nounroll
                    for (int k = 0; k < n; ++k) pair(std::false_type{});
                    pair(std::true_type{});
                }
                while (d2 >= 1) {  // whole blocks: two pairs per loop body (the streams' registers are back where they were: see the forward sweep)
                    block_head();
#pragma nounroll
                    for (int k = 0; k < RS_K / 4 - 1; ++k) pair(std::false_type{}), pair(std::false_type{});
                    pair(std::false_type{});
                    pair(std::true_type{});
                }
            }
            // total from the backward side: the lattice point (0, 0) is slot j0 of anti-diagonal 0
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (jr[r] == j0) {
                    const RCell cz = Q.A.c[r];
                    const float raw = rs_dot5(mdl->start + rs * 5, cz);
                    float bm = 0.f;
                    int be = E_DEAD;
                    if (raw > 0.f) {
                        int k;
                        bm = __builtin_frexpf(raw, &k);
                        be = Q.e + k;
                    }
                    reinterpret_cast<float *>(lmisc)[2] = bm;
                    lmisc[3] = be;
                }
            __syncthreads();
            out.btot_m = unif(reinterpret_cast<float *>(lmisc)[2]);
            out.btot_e = uni(lmisc[3]);
        }
        if (lane == 0) {
            out.npairs = cnt;
            if (cnt > pair_cap) out.status = NPR_ERR_CAPACITY;
            // one exponent per row may not have been enough -- also when nothing arrived at the end corner: the per-cell kernel decides
            // whether the band really carries no probability
            if (smax >= NPR_RS_S_LIMIT || !alive) out.status = TASK_RERUN;
            a.outs[t] = out;
        }
        int nt = 0;
        if (lane == 0) nt = atomicAdd(a.queue, 1);
        t = uni(nt) + static_cast<int>(gridDim.x);
    }
}


}  // namespace

size_t rs_lds_bytes() { return 0; }  // static LDS only

// sw: some loaded model has a short-gap switch (shortGapX <-> shortGapY); without one the two multiply-adds per cell and direction that would
// add an exact zero are not issued.  flat: every loaded model's gap emissions are exactly 2^-2 for every base (the shipped ones' are): they come
// from a select instead of the LDS tables.  Same bits either way (npr_rs.h).
template <int R>
static int launch_rs_r(const KernelArgs &a, bool sw, bool flat, int grid, hipStream_t s) {
    if (sw) hipLaunchKernelGGL((k_dp_rs<R, true, false>), dim3(grid), dim3(WAVE), 0, s, a);
    else if (flat) hipLaunchKernelGGL((k_dp_rs<R, false, true>), dim3(grid), dim3(WAVE), 0, s, a);
    else hipLaunchKernelGGL((k_dp_rs<R, false, false>), dim3(grid), dim3(WAVE), 0, s, a);
    return static_cast<int>(hipGetLastError());
}
int launch_rs(const KernelArgs &a, int R, int grid, void *stream, bool sw, bool flat) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (R == 1) return launch_rs_r<1>(a, sw, flat, grid, s);
    if (R == 2) return launch_rs_r<2>(a, sw, flat, grid, s);
    if (R == 4) return launch_rs_r<4>(a, sw, flat, grid, s);
    return static_cast<int>(hipErrorInvalidValue);
}

}  // namespace npr
