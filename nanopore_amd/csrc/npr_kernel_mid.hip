// npr_kernel_mid.hip -- k_dp_mid_rs<R>: k_dp_rs (npr_kernel_rs.hip) with a read's two sweeps on TWO wavefronts that MEET IN THE MIDDLE (round 5).
//
// One wavefront per read makes a read a serial chain of 2 D steps (D = lX + lY anti-diagonals): a launch lasts at least as long as its
// longest read, and a batch with fewer reads than the chip has wavefront slots leaves the rest idle (BASELINE.json configs[1]; one
// rank's eighth of configs[3] -- the per-read fan-out of nanopore/analyses/utils.py:565-570).  Here wavefront 0 sweeps FORWARD
// from anti-diagonal 0 to the cut row c (a multiple of RS_K near D / 2), storing its match rows as k_dp_rs does, while wavefront 1
// sweeps BACKWARD from D down to c, storing ITS match rows (rows c + 1 .. D of the same region: 4 bytes per cell, as many bytes as
// k_dp_rs moves).  Where they meet, every path of the lattice either passes through a cell of anti-diagonal c or jumps over it with
// a match move that lands on c + 1, so
//     total' = sum over row c of F_s B_s (five states)  +  sum over row c + 1 of F_match B_match
// -- wavefront 0 hands F(c) and F_match(c + 1) over through LDS, wavefront 1 forms the sum.  Then both go on: wavefront 0 forward from
// c + 1 to D against the backward rows wavefront 1 left, wavefront 1 backward from c to 0 against the forward rows, each turning its
// half of the lattice into posteriors as it goes: D steps of latency instead of 2 D, no third pass.
//
// SAME BITS as k_dp_rs.  The two sweeps ARE k_dp_rs's (same steps, same renormalising rows, same exponents), but k_dp_rs divides by
// the total its forward sweep arrives with at the end corner, which wavefront 0 only knows when it is done, and total' differs from
// it in the last bits.  So the second halves emit CANDIDATES: q' = F * ldexp(B, eF + eB - eTot') -- a power of two away from
// k_dp_rs's intermediate, exactly -- for every cell with q' / totMant' >= threshold * (1 - 2^-10); when both totals are known
// (and agree to 2^-12, else the task runs again per cell) one pass over the candidates (a hundredth of the cells) turns q' into
// k_dp_rs's p = ldexp(q', eTot' - eTot) * (1 / totMant), bit for bit, and drops the few that fall below the threshold after all.
// Wavefront 0 fills the task's pair list from the front and wavefront 1 from the back (no atomics in the sweeps); that pass closes
// the gap.  Range certificate, second pass per cell and outputs as k_dp_rs (npr_device.h).
#include <hip/hip_runtime.h>
#include <type_traits>

#include "npr_device.h"
#include "npr_frame.h"
#include "npr_rs.h"

namespace npr {

namespace {

constexpr int MID_REJ_CAP = 64;  // candidates of one task that may fall below the threshold after all (more: the task runs again per cell)

// candidates of one anti-diagonal: q = f * ldexp(b, s) where k_dp_rs forms p = (f * ldexp(b, s)) * inv_tot; BACK: slots from the end of the list
template <int R, bool BACK>
__device__ __forceinline__ void mid_emit(const PairSink &S, const float (&f)[R], const float (&b)[R], int d, int x0, int y0, const Masks<R> &mk, int s,
                                         float q_min, const int (&jr)[R], int &cnt) {
    float q[R];
    uint64_t hit[R], any = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        q[r] = f[r] * __builtin_ldexpf(b[r], s);
        hit[r] = __ballot(q[r] >= q_min) & mk.cell[r];  // (q_min = the lowered threshold times total's mantissa: the test needs no division)
        any |= hit[r];
    }
    if (d >= 2 && any) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (hit[r]) {
                const int before = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(hit[r] >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(hit[r]), 0));
                const int n = cnt + before;
                if (lanes_of(hit[r]) && n < S.cap) {
                    const uint32_t u = static_cast<uint32_t>(BACK ? S.cap - 1 - n : n) << 2;
                    rs_at<int32_t>(S.px, u) = x0 + jr[r] - 1 + S.xs;
                    rs_at<int32_t>(S.py, u) = y0 - jr[r] - 1 + S.ys;
                    rs_at<float>(S.pp, u) = q[r];
                }
                cnt += __popcll(hit[r]);
            }
        }
    }
}

// The two wavefronts of a workgroup meet: everything a wavefront has written (rows, exponents, LDS) is on its way before it passes.  (Each
// wavefront runs its own straight path through the task -- the barriers sit inside the paths, not at joins of them: a join would keep both
// paths' registers alive across it.)
__device__ __forceinline__ void mid_meet() {
    __builtin_amdgcn_s_waitcnt(0);
    __threadfence_block();
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

template <int R, bool SW, bool FLAT>
__global__ void __launch_bounds__(2 * WAVE) __attribute__((amdgpu_waves_per_eu(R == 1 ? 6 : (R == 2 ? NPR_MID_WAVES2 : 4)))) k_dp_mid_rs(KernelArgs a) {
    __shared__ __attribute__((aligned(16))) RsTables ltab_s;
    __shared__ __attribute__((aligned(16))) float lmodel[MODEL_FLOATS];
    // [0..1] total (forward, at the end corner), [2..3] total (backward), [4] sum of rebases, [5] / [12] candidates of wavefront 0 / 1,
    // [6] next task, [7] / [13] largest eF + eB of wavefront 0 / 1, [8] wavefront 0's exponent and [9] its x0 at the cut, [10..11] total', [14] rejects
    __shared__ int lmisc[16];
    __shared__ float xch[6][64 * R];
    __shared__ int lrej[MID_REJ_CAP];
    RsTables *ltab = &ltab_s;

    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = uni(static_cast<int>(threadIdx.x) >> 6);
    char *const F = a.F + uni64(a.region[blockIdx.x]) * 8;  // the workgroup's region: rows 0 .. c forward, c + 1 .. D backward; then the exponents
    const int voff = 4 * R * lane;
    int jr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) jr[r] = R * lane + r;

    int t = blockIdx.x, last_model = -1;
    while (t < a.ntasks) {
        const Task *tp = a.tasks + t;
        const int64_t x_off = uni64(tp->x_off), y_off = uni64(tp->y_off), ctl_off = uni64(tp->ctl_off), pair_off = uni64(tp->pair_off);
        const int lX = uni(tp->lX), lY = uni(tp->lY), D = uni(tp->D), pair_cap = min(uni(tp->pair_cap), (1 << 29) - 1) /* (pair slots are 32-bit byte offsets) */, flags = uni(tp->flags),
                  model = uni(tp->model), xs = uni(tp->xs), ys = uni(tp->ys);
        const int64_t half = rs_half_cells(static_cast<int64_t>(static_cast<uint32_t>(uni(tp->cells_pad))));
        int *const fexp = reinterpret_cast<int *>(F + 4 * half), *const bexp = fexp + (D / RS_K + 2);
        cptr32 ctl = (cptr32)(a.ctl + 2 * ctl_off);
        const __amdgpu_buffer_rsrc_t frs = rs_task_rsrc<R>(F);
        const int rs = flags & 1, re = (flags >> 1) & 1;
        const int c = (D / (2 * RS_K)) * RS_K;  // the cut row; D >= MID_MIN_D: c >= RS_K and D - c >= 2 RS_K

        __syncthreads();
        const bool new_model = model != last_model;  // (uniform.  The model and the tables made of it stay in LDS from task to task: a batch has one model, or a few)
        if (new_model) {
            const float *gm = reinterpret_cast<const float *>(a.models + model);
            for (int i = threadIdx.x; i < MODEL_FLOATS; i += 2 * WAVE) lmodel[i] = gm[i];
        }
        if (threadIdx.x < 16) lmisc[threadIdx.x] = threadIdx.x == 1 || threadIdx.x == 3 ? E_DEAD : (threadIdx.x == 7 || threadIdx.x == 13 ? -(1 << 30) : 0);
        __syncthreads();
        if (new_model) rs_build_tables(ltab, reinterpret_cast<const DevModel *>(lmodel), threadIdx.x, 2 * WAVE);
        last_model = model;
        StepEnv E;
        E.mdl = reinterpret_cast<const DevModel *>(lmodel);
        E.ltab = reinterpret_cast<const char *>(ltab);
        E.X = a.seq + x_off, E.Y = a.seq + y_off, E.lX = lX, E.lY = lY, E.lane = lane;
        const DevModel *mdl = E.mdl;
        const RowCtl<R> c0 = read_row_ctl<R>(ctl, 0);
        const int j0 = c0.jlo;  // slot of the lattice point (0, 0)
        // where the frame stands on the last anti-diagonal (the backward sweep starts there): an X-step into every odd one, a Y-step
        // into every even one, and the rebases of the schedule
        {
            int sum = 0;
            const uint32_t *gw1 = a.ctl + 2 * ctl_off;
            for (int dd = 1 + static_cast<int>(threadIdx.x); dd <= D; dd += 2 * WAVE) sum += ctl_rebase_of<R>(gw1[2 * dd + 1]);
            if (sum) atomicAdd(&lmisc[4], sum);
        }
        __syncthreads();
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
        const int rebs = uni(lmisc[4]);
        const int xD = -j0 + (D + 1) / 2 + rebs, yD = j0 + D / 2 - rebs;
        const bool fits = D >= MID_MIN_D;  // (the host sends shorter tasks to k_dp_rs; one that got here runs again per cell)

        // ---- the state of a wavefront's sweep, alive across the two meetings ----
        RsState<R> Q;
        Q.A = zero_rdiag<R>(), Q.B = zero_rdiag<R>();
        Q.e = 0, Q.x0 = 0, Q.y0 = 0;
        Q.S.xcap = RS_NX, Q.S.ycap = RS_N8;
        RFRow<R> ra, rb;  // the OTHER sweep's rows of the even / odd anti-diagonals, loaded one anti-diagonal ahead
#pragma unroll
        for (int r = 0; r < R; ++r) ra.v[r] = rb.v[r] = 0.f;
        int d = 1, cnt = 0, smax = -(1 << 30), sblk = 0, eo = 0;
        CtlPair w0{0u, 0u, 0u, 0u}, w1 = w0, w2 = w0;  // control words: the pair being computed, the next one, the one after (wavefront 1: w2 only)
        RowCtl<R> cur = c0, nxt = c0;
        uint32_t m1 = 0, m2 = 0;
        float q_min = 0.f;  // candidates: q' >= threshold * (1 - 2^-10) * totMant'
        int pte = 0;
        PairSink sink{a.px + pair_off, a.py + pair_off, a.pp + pair_off, 0, pair_cap, xs, ys, a.threshold * (1.0f - 0x1p-10f)};

        constexpr std::true_type Y{};
        constexpr std::false_type N{};
        // wavefront 0, one half of a pair of anti-diagonals: the X-step into the odd one / the Y-step into the even one; EMIT: against the backward rows
        auto xhalf = [&](auto emit, auto ahead, const CtlPair &w) __attribute__((always_inline)) {
            constexpr bool EMIT = decltype(emit)::value;
            const RowCtl<R> cx = row_ctl_of_words<R>(w.a0, w.a1);
            if constexpr (EMIT && decltype(ahead)::value) rs_load_row<R>(frs, ra, row_ctl_of_words<R>(w.b0, w.b1), voff);  // for the step after this one
            RS_FWD_REBASE(cx.reb);
            rs_fwd_x_step<R, false, SW, FLAT>(E, Q.B, Q.A, Q.S, Q.x0, cx.mk, cx.moved);
            if constexpr (EMIT) {
                float fv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) fv[r] = Q.B.c[r].m;
                mid_emit<R, false>(sink, fv, rb.v, d, Q.x0, Q.y0, cx.mk, sblk, q_min, jr, cnt);
            } else {
                rs_store_row<R>(frs, Q.B, cx, voff);
            }
        };
        auto yhalf = [&](auto emit, auto last, const CtlPair &w) __attribute__((always_inline)) {
            constexpr bool EMIT = decltype(emit)::value;
            const RowCtl<R> cy = row_ctl_of_words<R>(w.b0, w.b1);
            if constexpr (EMIT) {
                // (always issued, so that the compiler may wait for the older of two rows in flight: past the last row, this row's again)
                const bool more = d + 2 <= D;
                rs_load_row<R>(frs, rb, row_ctl_of_words<R>(more ? w1.a0 : w.b0, more ? w1.a1 : w.b1), voff);
            }
            RS_FWD_REBASE(cy.reb);
            rs_fwd_y_step<R, false, SW, FLAT>(E, Q.A, Q.B, Q.S, Q.y0, cy.mk, cy.moved);
            if constexpr (decltype(last)::value) {
                Q.e += rs_renorm<R>(Q.A, Q.B);
                if constexpr (EMIT) sblk = note_s(smax, Q.e + eo) - pte;
                else if (lane == 0) fexp[(d + 1) / RS_K] = Q.e;
            }
            if constexpr (EMIT) {
                float fv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) fv[r] = Q.A.c[r].m;
                mid_emit<R, false>(sink, fv, ra.v, d + 1, Q.x0, Q.y0, cy.mk, sblk, q_min, jr, cnt);
            } else {
                rs_store_row<R>(frs, Q.A, cy, voff);
            }
        };
        auto fpair = [&](auto emit, auto last) __attribute__((always_inline)) {
            w0 = w1, w1 = w2;
            w2 = ctl_scalar2(ctl, d + 4);  // (a few words past the task's last row at most: inside d_ctl or its padding)
            xhalf(emit, Y, w0);
            yhalf(emit, last, w0);
            d += 2;
        };
        auto fhead = [&](auto emit) __attribute__((always_inline)) {  // at the head of a block of RS_K anti-diagonals, d = 1 (mod RS_K)
            feed8_ahead<+1, RS_XS>(Q.S.fx, E.X, lX, Q.x0 + 64 * R - 1, lane);
            feed8_ahead<+1>(Q.S.fy, E.Y, lY, Q.y0, lane);
            if constexpr (decltype(emit)::value) {
                eo = ((cptr_i32)bexp)[(d + RS_K - 1) / RS_K];
                sblk = note_s(smax, Q.e + eo) - pte;
            }
        };
        // wavefront 1, a pair of anti-diagonals (d, d - 1), d odd: the undone Y-step into d + 1, then the undone X-step into d; EMIT: against the
        // forward rows
        auto bpair = [&](auto emit, auto last) __attribute__((always_inline)) {
            constexpr bool EMIT = decltype(emit)::value;
            const CtlPair q = w2;
            w2 = ctl_scalar2(ctl, d - 4);  // (down to row -3 of the task: kCtlFrontPad)
            int reb = cur.reb;
            cur = nxt;
            nxt = row_ctl_of_words<R>(q.b0, q.b1);
            if constexpr (EMIT) rs_load_row<R>(frs, ra, nxt, voff);  // for the step after this one
            RS_BWD_REBASE(reb);
            rs_bwd_y_step<R, false, SW, FLAT>(E, Q.B, Q.A, Q.S, Q.y0, cur.mk, m2);
            m2 = m1, m1 = cur.moved;
            if constexpr (EMIT) {
                float bv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) bv[r] = Q.B.c[r].m;
                mid_emit<R, true>(sink, rb.v, bv, d, Q.x0, Q.y0, cur.mk, sblk, q_min, jr, cnt);
            } else {
                rs_store_row<R>(frs, Q.B, cur, voff);
            }
            reb = cur.reb;
            cur = nxt;
            if (!decltype(last)::value || d >= 2) {
                nxt = row_ctl_of_words<R>(q.a0, q.a1);
                if constexpr (EMIT) rs_load_row<R>(frs, rb, nxt, voff);
            }
            RS_BWD_REBASE(reb);
            rs_bwd_x_step<R, false, SW, FLAT>(E, Q.A, Q.B, Q.S, Q.x0, cur.mk, m2);
            m2 = m1, m1 = cur.moved;
            if constexpr (decltype(last)::value) {
                Q.e += rs_renorm<R>(Q.A, Q.B);
                if constexpr (EMIT) sblk = note_s(smax, eo + Q.e) - pte;
                else if (lane == 0) bexp[(d - 1) / RS_K] = Q.e;
            }
            if constexpr (EMIT) {
                float bv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) bv[r] = Q.A.c[r].m;
                mid_emit<R, true>(sink, ra.v, bv, d - 1, Q.x0, Q.y0, cur.mk, sblk, q_min, jr, cnt);
            } else {
                // (not the cut row's: wavefront 0's forward row lies there.  A scalar test around the store alone: two variants of the pair behind
                // a test would meet at a join, which is paid with a second copy of the rows' registers)
                if (d - 1 != c) rs_store_row<R>(frs, Q.A, cur, voff);
            }
            d -= 2;
        };
        auto bhead = [&](auto emit) __attribute__((always_inline)) {
            feed8_ahead<-1, RS_XS>(Q.S.fx, E.X, lX, Q.x0 - 1, lane);
            feed8_ahead<-1>(Q.S.fy, E.Y, lY, Q.y0 - 64 * R, lane);
            if constexpr (decltype(emit)::value) {
                eo = ((cptr_i32)fexp)[d / RS_K];
                sblk = note_s(smax, eo + Q.e) - pte;
            }
        };
        // =============================== the two paths ===============================
        float ptm = 0.f;
        if (!fits) {
            mid_meet(), mid_meet(), mid_meet();
        } else if (wv == 0) {
            // ---- wavefront 0: forward, as k_dp_rs, rows 0 .. c stored; then the X-step into c + 1 ----
            Q.x0 = -j0, Q.y0 = j0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                Q.S.X.b[r] = base8<RS_XS>(E.X, lX, Q.x0 + jr[r] - 1);
                Q.S.Y.b[r] = base8(E.Y, lY, Q.y0 - jr[r] - 1);
            }
            feed8_init<+1, RS_XS>(Q.S.fx, E.X, lX, Q.x0 + 64 * R - 1, lane);  // first X-step injects X[(x0 + 1) + 64R - 2]
            feed8_init<+1>(Q.S.fy, E.Y, lY, Q.y0, lane);                       // first Y-step injects Y[(y0 + 1) - 1]
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (jr[r] == j0) {
                    RCell s0;
                    s0.m = mdl->start[rs * 5 + 0], s0.sx = mdl->start[rs * 5 + 1], s0.sy = mdl->start[rs * 5 + 2];
                    s0.lx = mdl->start[rs * 5 + 3], s0.ly = mdl->start[rs * 5 + 4];
                    Q.A.c[r] = s0;
                }
            if (lane == 0) fexp[0] = 0;
            rs_store_row<R>(frs, Q.A, c0, voff);
            w1 = ctl_scalar2(ctl, 1), w2 = ctl_scalar2(ctl, 3);
            while (d + RS_K - 1 <= c) {
                fhead(N);
#pragma nounroll
                for (int k = 0; k < RS_K / 4 - 1; ++k) fpair(N, N), fpair(N, N);
                fpair(N, N);
                fpair(N, Y);
            }
            // d = c + 1, the head of the block the cut runs through: its first X-step (the row is NOT stored: wavefront 1's row c + 1 lies there)
            fhead(N);
            w0 = w1, w1 = w2;
            w2 = ctl_scalar2(ctl, d + 4);
            {
                const RowCtl<R> cx = row_ctl_of_words<R>(w0.a0, w0.a1);
                RS_FWD_REBASE(cx.reb);
                rs_fwd_x_step<R, false, SW, FLAT>(E, Q.B, Q.A, Q.S, Q.x0, cx.mk, cx.moved);
            }
            // F(c), five states, and F_match(c + 1) by slot (wavefront 1 knows how the two frames lie to each other: [9])
#pragma unroll
            for (int r = 0; r < R; ++r) {
                xch[0][jr[r]] = Q.A.c[r].m, xch[1][jr[r]] = Q.A.c[r].sx, xch[2][jr[r]] = Q.A.c[r].sy, xch[3][jr[r]] = Q.A.c[r].lx, xch[4][jr[r]] = Q.A.c[r].ly;
                xch[5][jr[r]] = Q.B.c[r].m;
            }
            if (lane == 0) lmisc[8] = Q.e, lmisc[9] = Q.x0;
            mid_meet();  // rows 0 .. c, their exponents, F(c) and F_match(c + 1) are out
            mid_meet();  // wavefront 1 has formed the total at the cut
            __builtin_amdgcn_s_dcache_inv();  // wavefront 1's exponents come back through the scalar cache
            ptm = unif(reinterpret_cast<float *>(lmisc)[10]);
            pte = uni(lmisc[11]);
            if (ptm > 0.f) {
                q_min = sink.threshold * ptm;
                // rows c + 1 (held in B) and c + 2: the backward rows they pair with; then the Y-step that completes the pair
                fhead(Y);
                rs_load_row<R>(frs, rb, row_ctl_of_words<R>(w0.a0, w0.a1), voff);
                rs_load_row<R>(frs, ra, row_ctl_of_words<R>(w0.b0, w0.b1), voff);
                {
                    const RowCtl<R> cx = row_ctl_of_words<R>(w0.a0, w0.a1);
                    float fv[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) fv[r] = Q.B.c[r].m;
                    mid_emit<R, false>(sink, fv, rb.v, d, Q.x0, Q.y0, cx.mk, sblk, q_min, jr, cnt);
                }
                yhalf(Y, N, w0);
                d += 2;
                // the other seven pairs of that block, whole blocks, the rows after the last whole block
#pragma nounroll
                for (int k = 0; k < RS_K / 4 - 1; ++k) fpair(Y, N), fpair(Y, N);
                fpair(Y, Y);
                while (d + RS_K - 1 <= D) {
                    fhead(Y);
#pragma nounroll
                    for (int k = 0; k < RS_K / 4 - 1; ++k) fpair(Y, N), fpair(Y, N);
                    fpair(Y, N);
                    fpair(Y, Y);
                }
                if (d <= D) {  // the rows after the last whole block
                    fhead(Y);
#pragma nounroll
                    while (d + 1 <= D) fpair(Y, N);
                    if (d <= D) xhalf(Y, N, w1);  // D odd: one more X-step
                }
                // total probability at the end corner (lX, lY): slot lX - x0 of the last anti-diagonal -- the total k_dp_rs divides by
                const int je = lX - Q.x0;
                const bool oddD = D & 1;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (jr[r] == je) {
                        const RCell ce = oddD ? Q.B.c[r] : Q.A.c[r];
                        const float raw = rs_dot5(mdl->end + re * 5, ce);
                        if (raw > 0.f) {
                            int k;
                            reinterpret_cast<float *>(lmisc)[0] = __builtin_frexpf(raw, &k);
                            lmisc[1] = Q.e + k;
                        }
                    }
                if (lane == 0) lmisc[5] = cnt, lmisc[7] = smax;
            }
            mid_meet();
        } else {
            // ---- wavefront 1: backward, as k_dp_rs's, rows D .. c + 1 stored with their exponents: rows RS_K (k - 1) + 1 .. RS_K k share bexp[k] ----
            Q.x0 = xD, Q.y0 = yD;
            const bool oddD = D & 1;
            cur = read_row_ctl<R>(ctl, D);
            m1 = cur.moved, m2 = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                Q.S.X.b[r] = base8<RS_XS>(E.X, lX, Q.x0 + jr[r]);
                Q.S.Y.b[r] = base8(E.Y, lY, Q.y0 - jr[r]);
                if (Q.x0 + jr[r] == lX) {  // the end corner (it is in the band by construction)
                    RCell e0;
                    e0.m = mdl->end[re * 5 + 0], e0.sx = mdl->end[re * 5 + 1], e0.sy = mdl->end[re * 5 + 2];
                    e0.lx = mdl->end[re * 5 + 3], e0.ly = mdl->end[re * 5 + 4];
                    if (oddD) Q.B.c[r] = e0; else Q.A.c[r] = e0;
                }
            }
            // first undone X-step injects X[x0 - 1] at slot 0; first undone Y-step injects Y[y0 - 64R] on top
            feed8_init<-1, RS_XS>(Q.S.fx, E.X, lX, Q.x0 - 1, lane);
            feed8_init<-1>(Q.S.fy, E.Y, lY, Q.y0 - 64 * R, lane);
            if (lane == 0) bexp[(D + RS_K - 1) / RS_K] = 0;
            rs_store_row<R>(frs, oddD ? Q.B : Q.A, cur, voff);
            nxt = read_row_ctl<R>(ctl, D - 1);
            d = D - 1;
            if (oddD) {  // peel one even anti-diagonal so that the pairs below always start on an odd one
                const int reb = cur.reb;
                cur = nxt;
                nxt = read_row_ctl<R>(ctl, d - 1);
                RS_BWD_REBASE(reb);
                rs_bwd_x_step<R, true, SW, FLAT>(E, Q.A, Q.B, Q.S, Q.x0, cur.mk, m2);
                m2 = m1, m1 = cur.moved;
                if ((d & (RS_K - 1)) == 0) {
                    Q.e += rs_renorm<R>(Q.A, Q.B);
                    if (lane == 0) bexp[d / RS_K] = Q.e;
                }
                rs_store_row<R>(frs, Q.A, cur, voff);  // (d > c: D - c >= 2 RS_K)
                d -= 1;
            }
            w2 = ctl_scalar2(ctl, d - 2);
            {  // blocks that end on a renormalising row; the first as long as it takes to get to one (d > c: D - c >= 2 RS_K)
                bhead(N);
                const int n = (d & (RS_K - 1)) >> 1;
#pragma nounroll
                for (int k = 0; k < n; ++k) bpair(N, N);
                bpair(N, Y);
            }
            while (d > c) {  // whole blocks: two pairs per loop body (the streams' registers are back where they were: k_dp_rs)
                bhead(N);
#pragma nounroll
                for (int k = 0; k < RS_K / 4 - 1; ++k) bpair(N, N), bpair(N, N);
                bpair(N, N);
                bpair(N, Y);
            }
            // d = c - 1; Q.A holds row c, Q.B row c + 1; cur the words of c, nxt those of c - 1
            mid_meet();
            {
            // ---- the total at the cut ----
            // The frames: wavefront 0 stands on c + 1 = the frame of c, rebased by r, one X-step on; its row c has moved with the rebase.  Slot j of
            // either row here is slot j - r there.  Both factors scaled by 2^-40: a cell that matters has F B >= 2^-(NPR_RS_S_LIMIT) or so below 2^182.
            const int r_ = uni(lmisc[9]) - Q.x0 - 1;
            float acc = 0.f;
            constexpr float SC = 0x1p-40f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = jr[r] - r_;
                const bool ok = i >= 0 && i < 64 * R;
                const int ii = ok ? i : 0;
                const float f0 = ok ? xch[0][ii] : 0.f, f1 = ok ? xch[1][ii] : 0.f, f2 = ok ? xch[2][ii] : 0.f, f3 = ok ? xch[3][ii] : 0.f, f4 = ok ? xch[4][ii] : 0.f,
                            g = ok ? xch[5][ii] : 0.f;
                acc = __builtin_fmaf(f0 * SC, Q.A.c[r].m * SC, acc);
                acc = __builtin_fmaf(f1 * SC, Q.A.c[r].sx * SC, acc);
                acc = __builtin_fmaf(f2 * SC, Q.A.c[r].sy * SC, acc);
                acc = __builtin_fmaf(f3 * SC, Q.A.c[r].lx * SC, acc);
                acc = __builtin_fmaf(f4 * SC, Q.A.c[r].ly * SC, acc);
                acc = __builtin_fmaf(g * SC, Q.B.c[r].m * SC, acc);
            }
            const float tot = unif(wave_sum_f32(acc));
            if (lane == 0) {
                int k = 0;
                const float tm = (tot > 0.f && tot < __builtin_inff()) ? __builtin_frexpf(tot, &k) : 0.f;
                reinterpret_cast<float *>(lmisc)[10] = tm;
                lmisc[11] = uni(lmisc[8]) + Q.e + 80 + k;
            }
            }
            mid_meet();
            __builtin_amdgcn_s_dcache_inv();  // wavefront 0's exponents come back through the scalar cache
            ptm = unif(reinterpret_cast<float *>(lmisc)[10]);
            pte = uni(lmisc[11]);
            if (ptm > 0.f) {
                q_min = sink.threshold * ptm;
                // row c against its forward row, then k_dp_rs's blocks from c - 1 down
                rs_load_row<R>(frs, ra, cur, voff);
                rs_load_row<R>(frs, rb, nxt, voff);
                eo = ((cptr_i32)fexp)[c / RS_K];
                sblk = note_s(smax, eo + Q.e) - pte;
                {
                    float bv[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) bv[r] = Q.A.c[r].m;
                    mid_emit<R, true>(sink, ra.v, bv, c, Q.x0, Q.y0, cur.mk, sblk, q_min, jr, cnt);
                }
                while (d >= 1) {  // whole blocks: d = c - 1 = RS_K m - 1
                    bhead(Y);
#pragma nounroll
                    for (int k = 0; k < RS_K / 4 - 1; ++k) bpair(Y, N), bpair(Y, N);
                    bpair(Y, N);
                    bpair(Y, Y);
                }
                // total from the backward side: the lattice point (0, 0) is slot j0 of anti-diagonal 0
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (jr[r] == j0) {
                        const float raw = rs_dot5(mdl->start + rs * 5, Q.A.c[r]);
                        if (raw > 0.f) {
                            int k;
                            reinterpret_cast<float *>(lmisc)[2] = __builtin_frexpf(raw, &k);
                            lmisc[3] = Q.e + k;
                        }
                    }
                if (lane == 0) lmisc[12] = cnt, lmisc[13] = smax;
            }
            mid_meet();
        }
        ptm = unif(reinterpret_cast<float *>(lmisc)[10]);
        pte = uni(lmisc[11]);
        const bool alive1 = fits && ptm > 0.f;

        // =============================== both totals known: candidates -> k_dp_rs's pairs ===============================
        const float tot_m = unif(reinterpret_cast<float *>(lmisc)[0]);
        const int tot_e = uni(lmisc[1]);
        TaskOut out;
        out.tot_m = tot_m, out.tot_e = tot_e, out.btot_m = unif(reinterpret_cast<float *>(lmisc)[2]), out.btot_e = uni(lmisc[3]);
        out.npairs = 0, out.status = NPR_OK;
        const int nA = uni(lmisc[5]), nB = uni(lmisc[12]);
        const int de = pte - tot_e;
        bool good = alive1 && tot_m > 0.f && de >= -1 && de <= 1;
        int why = !alive1 ? 1 : (!(tot_m > 0.f) ? 2 : (!good ? 3 : 0));
        float ratio = 0.f;
        if (good) {
            ratio = __builtin_ldexpf(ptm, de) / tot_m;  // total' / total
            good = ratio >= 1.0f - 0x1p-12f && ratio <= 1.0f + 0x1p-12f;
            if (!good) why = 4;
        }
        if (good && max(uni(lmisc[7]), uni(lmisc[13])) - tot_e >= NPR_RS_S_LIMIT) good = false, why = 5;  // one exponent per row may not have been enough
        if (!good) {
            out.status = TASK_RERUN, out.btot_m = ratio, out.btot_e = de, out.npairs = why;  // (also when nothing arrives at the cut: the per-cell kernel says whether the band carries no probability)
        } else if (nA + nB > pair_cap) {
            // The two halves of the list have run into each other.  The read is reported and run again with a larger list, but whoever looks at
            // this one meanwhile (the MEA stage takes every task's first pair_cap entries) must find posteriors in it, not candidates: the
            // coordinates are some cell's either way (the two wavefronts' cells are disjoint: no pair twice), the values are made harmless.
            out.npairs = nA + nB, out.status = NPR_ERR_CAPACITY;
            float *const pp = a.pp + pair_off;
            for (int i = threadIdx.x; i < pair_cap; i += 2 * WAVE) pp[i] = a.threshold;
        } else {
            const float inv = 1.0f / tot_m, thr = a.threshold;
            int32_t *const px = a.px + pair_off, *const py = a.py + pair_off;
            float *const pp = a.pp + pair_off;
            const int tid = threadIdx.x;
            for (int i = tid; i < nA; i += 2 * WAVE) {
                const float p = __builtin_ldexpf(pp[i], de) * inv;
                pp[i] = p;
                if (!(p >= thr)) {
                    const int k = atomicAdd(&lmisc[14], 1);
                    if (k < MID_REJ_CAP) lrej[k] = i;
                }
            }
            // wavefront 1's, from the lowest address up: a block moved towards lower addresses, 128 entries read by BOTH wavefronts, then written by both.
            // The two meet before and after every block's writes whatever the free gap between the halves: the wavefronts are not synchronised when
            // they get here (the loop above has divergent atomics), and with a gap below nB a wavefront one block ahead would write what the other has
            // not read yet (round 5 took the barriers only for gaps below 128 entries: ADVICE r5).  The pass covers a hundredth of the cells.
            for (int base = 0; base < nB; base += 2 * WAVE) {
                const int k = base + tid;
                int x = 0, y = 0;
                float q = 0.f;
                if (k < nB) x = px[pair_cap - nB + k], y = py[pair_cap - nB + k], q = pp[pair_cap - nB + k];
                __syncthreads();
                if (k < nB) {
                    const float p = __builtin_ldexpf(q, de) * inv;
                    px[nA + k] = x, py[nA + k] = y, pp[nA + k] = p;
                    if (!(p >= thr)) {
                        const int j = atomicAdd(&lmisc[14], 1);
                        if (j < MID_REJ_CAP) lrej[j] = nA + k;
                    }
                }
                __syncthreads();
            }
            __syncthreads();
            const int nrej = uni(lmisc[14]);
            out.npairs = nA + nB - nrej;
            if (nrej > MID_REJ_CAP) {
                out.status = TASK_RERUN, out.npairs = 0, out.btot_m = 0.f, out.btot_e = E_DEAD;
            } else if (nrej > 0 && tid == 0) {
                // the candidates that fell below the threshold after all, from the highest index down: each takes the list's last entry
                for (int i = 1; i < nrej; ++i) {
                    const int v = lrej[i];
                    int j = i - 1;
                    for (; j >= 0 && lrej[j] < v; --j) lrej[j + 1] = lrej[j];
                    lrej[j + 1] = v;
                }
                int n = nA + nB;
                for (int i = 0; i < nrej; ++i) {
                    const int at = lrej[i];
                    --n;
                    if (at != n) px[at] = px[n], py[at] = py[n], pp[at] = pp[n];
                }
            }
        }
        if (threadIdx.x == 0) {
            a.outs[t] = out;
            lmisc[6] = atomicAdd(a.queue, 1);
        }
        __syncthreads();
        t = uni(lmisc[6]) + static_cast<int>(gridDim.x);
    }
}

}  // namespace

// sw: some loaded model has a short-gap switch (shortGapX <-> shortGapY); without one the two multiply-adds per cell and direction that would
// add an exact zero are not issued.  flat: every loaded model's gap emissions are exactly 2^-2 for every base (the shipped ones' are): they come
// from a select instead of the LDS tables.  Same bits either way (npr_rs.h).
template <int R>
static int launch_mid_rs_r(const KernelArgs &a, bool sw, bool flat, int grid, hipStream_t s) {
    if (sw) hipLaunchKernelGGL((k_dp_mid_rs<R, true, false>), dim3(grid), dim3(2 * WAVE), 0, s, a);
    else if (flat) hipLaunchKernelGGL((k_dp_mid_rs<R, false, true>), dim3(grid), dim3(2 * WAVE), 0, s, a);
    else hipLaunchKernelGGL((k_dp_mid_rs<R, false, false>), dim3(grid), dim3(2 * WAVE), 0, s, a);
    return static_cast<int>(hipGetLastError());
}
int launch_mid_rs(const KernelArgs &a, int R, int grid, void *stream, bool sw, bool flat) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (R == 1) return launch_mid_rs_r<1>(a, sw, flat, grid, s);
    if (R == 2) return launch_mid_rs_r<2>(a, sw, flat, grid, s);
    if (R == 4) return launch_mid_rs_r<4>(a, sw, flat, grid, s);
    return static_cast<int>(hipErrorInvalidValue);
}

}  // namespace npr
