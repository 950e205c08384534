// npr_kernel_stair.hip -- k_dp_stair<R>: the register-resident ("systolic") DP kernel for narrow bands.
//
// Same recurrences and the same per-cell arithmetic (npr_cell.h) as k_dp_generic -- cactus_realign's banded
// five-state forward / backward / posterior pass, SURVEY.md 8a rows a5.3-a5.5, reference call sites
// nanopore/analyses/utils.py:587, alignmentUncertainty.py:41, marginAlignSnpCaller.py:136-146 -- for bands of fewer
// than 64*R cells per anti-diagonal whose edges move by one cell per anti-diagonal: every fixed-width configuration
// (BASELINE.json "band=100/200") and every anchor stripe.
//
// Mapping (one read per 64-lane wavefront, no LDS traffic in the recurrence, no MFMA):
//   * the wavefront holds a FRAME of 64*R lattice points of the current anti-diagonal, slot j = (x0 + j, y0 - j),
//     slot j in lane j / R, register j % R (blocked).  The frame advances by an X-step (x0 += 1) into every odd
//     anti-diagonal and a Y-step (y0 += 1) into every even one, so the (x-1, y-1) predecessor always sits in the same
//     slot, the loop body is straight-line code and nothing but the one crossing neighbour ever moves.  The band floats
//     inside the frame (first slot jlo, n cells), selected by wave-uniform lane masks built on the scalar unit; when it
//     drifts to a frame edge the frame schedule (stair_step in npr_sched.h, run by k_plan_sched at staging) asks for a REBASE: the whole
//     register state moves one slot, in place (about one anti-diagonal in twenty on noisy reads; every other one
//     inside a long gap);
//   * of the R neighbours on d-1 only ONE per state crosses a lane boundary: a single DPP wave_shl:1 / wave_shr:1;
//   * the two previous anti-diagonals stay in VGPRs (6*R registers each) and swap roles every step (the loops are
//     unrolled by two), so no anti-diagonal is ever copied;
//   * the reference streams through the wavefront towards lower lanes on X-steps and the read towards higher lanes
//     on Y-steps (one DPP move + one v_readlane injection per step), fed by 64-base blocks prefetched a block
//     ahead: no per-cell sequence loads;
//   * HMM tables in LDS (emission look-ups), transitions in VGPRs (R = 1) or SGPRs;
//   * forward match-state values stream to the wavefront's HBM scratch as one 8R-byte buffer store per lane and
//     stream back one anti-diagonal ahead of use in the backward sweep;
//   * the per-anti-diagonal control words are read through the scalar cache (constant address space), one ahead.
#include <hip/hip_runtime.h>

#include "npr_cell.h"
#include "npr_device.h"
#include "npr_frame.h"

namespace npr {

namespace {


// R = 2 is held at 80 VGPRs (6 wavefronts per SIMD; 3 spilled registers) -- measured 4 % faster than 85 VGPRs / 5 waves
template <int R>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(R == 2 ? 6 : 1))) k_dp_stair(KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lmodel = reinterpret_cast<float *>(smem);
    int *lmisc = reinterpret_cast<int *>(lmodel + MODEL_FLOATS);  // 8 ints: cell hand-off

    const int lane = threadIdx.x;
    // the wavefront's forward scratch: its own region (big realign batches: sized by the region's first task, a.region) or
    // the blockIdx-th uniform one
    int64_t fcell = static_cast<int64_t>(a.slot_base + blockIdx.x) * a.slot_stride;
    if (a.region) fcell = a.region[blockIdx.x];
    char *const F = a.F + uni64(fcell) * 8;
    const int voff = 8 * R * lane;  // byte offset of this lane's cells inside a row that starts at lane 0
    int jr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) jr[r] = R * lane + r;

    int t = blockIdx.x;
    while (t < a.ntasks) {
        const Task *tp = a.tasks + t;
        const int64_t x_off = uni64(tp->x_off), y_off = uni64(tp->y_off), ctl_off = uni64(tp->ctl_off),
                      pair_off = uni64(tp->pair_off);
        const int lX = uni(tp->lX), lY = uni(tp->lY), D = uni(tp->D), pair_cap = uni(tp->pair_cap),
                  flags = uni(tp->flags), model = uni(tp->model), xs = uni(tp->xs), ys = uni(tp->ys);
        // control words through the scalar cache: the array is never written by the kernel
        cptr32 ctl = (cptr32)(a.ctl + 2 * ctl_off);
        const __amdgpu_buffer_rsrc_t frs = task_rsrc<R>(F);
        const int rs = flags & 1, re = (flags >> 1) & 1;

        __syncthreads();
        {
            const float *gm = reinterpret_cast<const float *>(a.models + model);
            for (int i = lane; i < MODEL_FLOATS; i += WAVE) lmodel[i] = gm[i];
        }
        __syncthreads();
        StepEnv E;
        E.mdl = reinterpret_cast<const DevModel *>(lmodel);
        E.ltab = reinterpret_cast<const char *>(lmodel);
        E.X = a.seq + x_off, E.Y = a.seq + y_off, E.lX = lX, E.lY = lY, E.lane = lane;
        {
            // A VALU op with an SGPR source issues ~1.7x slower on gfx950 (tools/valu_rates: v_fma_f32 v,s,v,v
            // 4.7 vs 2.7 cycles), so with one cell per lane the 15 transitions stay in VGPRs.  With more cells
            // per lane the 15 registers would cost a wave of occupancy per SIMD, which costs more: SGPRs there.
            Trans tr = load_trans(E.mdl->T);
            if constexpr (R >= NPR_T_SGPR_MIN_R) {
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
        Diag<R> A = dead_diag<R>(), B = dead_diag<R>();
        Streams<R> S;  // X[x-1]*4 and Y[y-1]*4 of every slot
        const RowCtl<R> c0 = read_row_ctl<R>(ctl, 0);
        const int j0 = c0.jlo;  // slot of the lattice point (0, 0)
        int x0 = -j0, y0 = j0;  // lattice point of slot 0
#pragma unroll
        for (int r = 0; r < R; ++r) {
            S.X.b[r] = base4(E.X, lX, x0 + jr[r] - 1);
            S.Y.b[r] = base4(E.Y, lY, y0 - jr[r] - 1);
        }
        S.xcap = S.ycap = 16;
        feed_init<+1>(S.fx, E.X, lX, x0 + 64 * R - 1, lane);  // first X-step injects X[(x0 + 1) + 64R - 2]
        feed_init<+1>(S.fy, E.Y, lY, y0, lane);               // first Y-step injects Y[(y0 + 1) - 1]
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (jr[r] == j0) {
                Cell c;
                c.m = mdl->start[rs * 5 + 0], c.sx = mdl->start[rs * 5 + 1], c.sy = mdl->start[rs * 5 + 2];
                c.lx = mdl->start[rs * 5 + 3], c.ly = mdl->start[rs * 5 + 4];
                normalise(c, 0);
                A.c[r] = c;
            }
        store_row<R>(frs, A, c0, voff);
        RowCtl<R> nx = c0;
        if (D >= 1) nx = read_row_ctl<R>(ctl, 1);
        int d = 1;
        cptr32 cp = ctl + 2;  // the control words of d, walked by pointer: two scalar adds per pair of anti-diagonals
        for (; d + 1 <= D; d += 2, cp += 4) {
            RowCtl<R> cur = nx;
            nx = read_row_ctl_at<R>(cp + 2);  // one ahead
            if (cur.reb) fwd_rebase<R>(E, cur.reb, A, B, S, x0, y0);
            fwd_x_step<R>(d, E, B, A, S, x0, cur.mk);
            store_row<R>(frs, B, cur, voff);
            cur = nx;
            if (d + 2 <= D) nx = read_row_ctl_at<R>(cp + 4);
            if (cur.reb) fwd_rebase<R>(E, cur.reb, A, B, S, x0, y0);
            fwd_y_step<R>(d + 1, E, A, B, S, y0, cur.mk);
            store_row<R>(frs, A, cur, voff);
        }
        if (d <= D) {  // D odd: one more X-step, into B
            if (nx.reb) fwd_rebase<R>(E, nx.reb, A, B, S, x0, y0);
            fwd_x_step<R>(d, E, B, A, S, x0, nx.mk);
            store_row<R>(frs, B, nx, voff);
        }
        // total probability at the end corner (lX, lY): slot lX - x0 of the last anti-diagonal
        {
            const int je = lX - x0;
            const bool oddD = D & 1;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (jr[r] == je) {
                    const Cell c = oddD ? B.c[r] : A.c[r];
                    const float raw = dot5(mdl->end + re * 5, c);
                    float tm = 0.f;
                    int te = E_DEAD;
                    if (raw > 0.f) {
                        int k;
                        tm = __builtin_frexpf(raw, &k);
                        te = c.e + k;
                    }
                    reinterpret_cast<float *>(lmisc)[0] = tm;
                    lmisc[1] = te;
                }
        }
        __syncthreads();
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
        if (alive) {
            const float inv_tot = 1.0f / tot_m;
            const PairSink sink{a.px, a.py, a.pp, pair_off, pair_cap, xs, ys, a.threshold};
            // A holds the even anti-diagonals again, B the odd ones; fa / fb the forward rows that pair with them,
            // loaded one anti-diagonal ahead.  S now holds X[x]*4 and Y[y]*4 of every slot.
            A = dead_diag<R>(), B = dead_diag<R>();
            const bool oddD = D & 1;
            RowCtl<R> cur = read_row_ctl<R>(ctl, D);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                S.X.b[r] = base4(E.X, lX, x0 + jr[r]);
                S.Y.b[r] = base4(E.Y, lY, y0 - jr[r]);
                if (x0 + jr[r] == lX) {  // the end corner (it is in the band by construction)
                    Cell c;
                    c.m = mdl->end[re * 5 + 0], c.sx = mdl->end[re * 5 + 1], c.sy = mdl->end[re * 5 + 2];
                    c.lx = mdl->end[re * 5 + 3], c.ly = mdl->end[re * 5 + 4];
                    normalise(c, 0);
                    if (oddD) B.c[r] = c; else A.c[r] = c;
                }
            }
            S.xcap = S.ycap = 16;
            // first undone X-step injects X[x0 - 1] at slot 0; first undone Y-step injects Y[y0 - 64R] on top
            feed_init<-1>(S.fx, E.X, lX, x0 - 1, lane);
            feed_init<-1>(S.fy, E.Y, lY, y0 - 64 * R, lane);
            FRow<R> fa, fb;
#pragma unroll
            for (int r = 0; r < R; ++r) fa.v[r] = fb.v[r] = 0.f, fa.e[r] = fb.e[r] = E_DEAD;
            RowCtl<R> nxt = cur;
            if (oddD) {
                load_row<R>(frs, fb, cur, voff);
                nxt = read_row_ctl<R>(ctl, D - 1);
                load_row<R>(frs, fa, nxt, voff);
                emit_pairs<R>(sink, B, fb, D, x0, y0, cur.mk, tot_e, inv_tot, jr, cnt);
            } else {
                load_row<R>(frs, fa, cur, voff);
                if (D >= 1) {
                    nxt = read_row_ctl<R>(ctl, D - 1);
                    load_row<R>(frs, fb, nxt, voff);
                }
                emit_pairs<R>(sink, A, fa, D, x0, y0, cur.mk, tot_e, inv_tot, jr, cnt);
            }
            // `cur` is the control word of the anti-diagonal above the one computed next: its rebase is undone first
            int d2 = D - 1;
            if (oddD) {  // peel one even anti-diagonal so that the loop below always starts on an odd one
                const int reb = cur.reb;
                cur = nxt;
                if (d2 >= 1) {
                    nxt = read_row_ctl<R>(ctl, d2 - 1);
                    load_row<R>(frs, fb, nxt, voff);
                }
                if (reb) bwd_rebase<R>(E, reb, A, B, S, x0, y0);
                bwd_x_step<R>(d2, E, A, B, S, x0, cur.mk);
                emit_pairs<R>(sink, A, fa, d2, x0, y0, cur.mk, tot_e, inv_tot, jr, cnt);
                d2 -= 1;
            }
            cptr32 cq = ctl + 2 * static_cast<int64_t>(d2 - 2);  // the control words of d2 - 2 (read only while d2 >= 2)
            for (; d2 >= 1; d2 -= 2, cq -= 4) {  // d2 odd: undo the Y-step into d2 + 1, then the X-step into d2
                int reb = cur.reb;
                cur = nxt;
                nxt = read_row_ctl_at<R>(cq + 2);
                load_row<R>(frs, fa, nxt, voff);  // for the step after this one
                if (reb) bwd_rebase<R>(E, reb, A, B, S, x0, y0);
                bwd_y_step<R>(d2, E, B, A, S, y0, cur.mk);
                emit_pairs<R>(sink, B, fb, d2, x0, y0, cur.mk, tot_e, inv_tot, jr, cnt);
                reb = cur.reb;
                cur = nxt;
                if (d2 >= 2) {
                    nxt = read_row_ctl_at<R>(cq);
                    load_row<R>(frs, fb, nxt, voff);
                }
                if (reb) bwd_rebase<R>(E, reb, A, B, S, x0, y0);
                bwd_x_step<R>(d2 - 1, E, A, B, S, x0, cur.mk);
                emit_pairs<R>(sink, A, fa, d2 - 1, x0, y0, cur.mk, tot_e, inv_tot, jr, cnt);
            }
            // total from the backward side: the lattice point (0, 0) is slot j0 of anti-diagonal 0
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (jr[r] == j0) {
                    const Cell cz = A.c[r];
                    const float raw = dot5(mdl->start + rs * 5, cz);
                    float bm = 0.f;
                    int be = E_DEAD;
                    if (raw > 0.f) {
                        int k;
                        bm = __builtin_frexpf(raw, &k);
                        be = cz.e + k;
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
            a.outs[t] = out;
        }
        int nt = 0;
        if (lane == 0) nt = atomicAdd(a.queue, 1);
        t = uni(nt) + static_cast<int>(gridDim.x);
    }
}

// =====================================================================================================================
// k_dp_wide<R, NW>: the same register-resident sweep for bands too wide for one wavefront.  A workgroup of NW wavefronts
// holds ONE frame of NW*64*R slots, wavefront w the slots [w*64R, (w+1)*64R).  What changes against k_dp_stair:
//   * the neighbour that crosses a wavefront boundary comes through LDS: after every anti-diagonal each wavefront
//     publishes its two edge cells (6 values each) and the workgroup takes ONE barrier; the DPP move's `old` operand
//     (the edge lane's value) is then the neighbour wavefront's edge instead of the dead cell;
//   * every wavefront runs its own base streams (its window of the sequences is just offset);
//   * wavefronts whose slots are more than four slots away from the band skip the step (they still take the
//     barrier): the unanchored diamonds of the reference's own band (anchors +- diagonalExpansion, up to
//     splitMatrixBiggerThanThis = 3000 cells across) alternate with 21-cell stripes, and a stripe costs one wavefront;
//     a wavefront that the band re-enters rebuilds its streams from memory and restarts from dead cells;
//   * posterior slots are claimed from one LDS counter (their order is restored by the host's sort).
// Same cell arithmetic, same frame schedule, same forward-row layout: bit-identical results.
// =====================================================================================================================
__device__ __forceinline__ int dpp_from_above_f(float v, float edge) {
    return __builtin_amdgcn_update_dpp(fbits(edge), fbits(v), 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_from_below_f(float v, float edge) {
    return __builtin_amdgcn_update_dpp(fbits(edge), fbits(v), 0x138, 0xf, 0xf, false);
}
template <int R>
__device__ __forceinline__ Diag<R> shift_up(const Diag<R> &in, const Cell &edge) {
    Diag<R> o;
#pragma unroll
    for (int r = 0; r + 1 < R; ++r) o.c[r] = in.c[r + 1];
    o.c[R - 1].m = bitsf(dpp_from_above_f(in.c[0].m, edge.m));
    o.c[R - 1].sx = bitsf(dpp_from_above_f(in.c[0].sx, edge.sx));
    o.c[R - 1].sy = bitsf(dpp_from_above_f(in.c[0].sy, edge.sy));
    o.c[R - 1].lx = bitsf(dpp_from_above_f(in.c[0].lx, edge.lx));
    o.c[R - 1].ly = bitsf(dpp_from_above_f(in.c[0].ly, edge.ly));
    o.c[R - 1].e = dpp_from_above(in.c[0].e, edge.e);
    return o;
}
template <int R>
__device__ __forceinline__ Diag<R> shift_down(const Diag<R> &in, const Cell &edge) {
    Diag<R> o;
#pragma unroll
    for (int r = 1; r < R; ++r) o.c[r] = in.c[r - 1];
    o.c[0].m = bitsf(dpp_from_below_f(in.c[R - 1].m, edge.m));
    o.c[0].sx = bitsf(dpp_from_below_f(in.c[R - 1].sx, edge.sx));
    o.c[0].sy = bitsf(dpp_from_below_f(in.c[R - 1].sy, edge.sy));
    o.c[0].lx = bitsf(dpp_from_below_f(in.c[R - 1].lx, edge.lx));
    o.c[0].ly = bitsf(dpp_from_below_f(in.c[R - 1].ly, edge.ly));
    o.c[0].e = dpp_from_below(in.c[R - 1].e, edge.e);
    return o;
}
// in-place one-slot moves with the neighbour wavefront's edge cell (uniform) written into the vacated lane
__device__ __forceinline__ void dpp_up_inplace_f(float &v, float edge) {
    const int eb = uni(fbits(edge));
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_writelane_b32 %0, %1, 63" : "+v"(v) : "s"(eb));
}
__device__ __forceinline__ void dpp_down_inplace_f(float &v, float edge) {
    const int eb = uni(fbits(edge));
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_writelane_b32 %0, %1, 0" : "+v"(v) : "s"(eb));
}
template <int R>
__device__ __forceinline__ void diag_up_inplace(Diag<R> &g, const Cell &edge) {
#pragma unroll
    for (int r = 0; r + 1 < R; ++r) {
        rot_up(g.c[r].m, g.c[r + 1].m), rot_up(g.c[r].sx, g.c[r + 1].sx), rot_up(g.c[r].sy, g.c[r + 1].sy);
        rot_up(g.c[r].lx, g.c[r + 1].lx), rot_up(g.c[r].ly, g.c[r + 1].ly), rot_up(g.c[r].e, g.c[r + 1].e);
    }
    Cell &t = g.c[R - 1];
    dpp_up_inplace_f(t.m, edge.m), dpp_up_inplace_f(t.sx, edge.sx), dpp_up_inplace_f(t.sy, edge.sy);
    dpp_up_inplace_f(t.lx, edge.lx), dpp_up_inplace_f(t.ly, edge.ly);
    dpp_up_inplace(t.e, uni(edge.e));
}
template <int R>
__device__ __forceinline__ void diag_down_inplace(Diag<R> &g, const Cell &edge) {
#pragma unroll
    for (int r = R - 1; r > 0; --r) {
        rot_up(g.c[r].m, g.c[r - 1].m), rot_up(g.c[r].sx, g.c[r - 1].sx), rot_up(g.c[r].sy, g.c[r - 1].sy);
        rot_up(g.c[r].lx, g.c[r - 1].lx), rot_up(g.c[r].ly, g.c[r - 1].ly), rot_up(g.c[r].e, g.c[r - 1].e);
    }
    Cell &t = g.c[0];
    dpp_down_inplace_f(t.m, edge.m), dpp_down_inplace_f(t.sx, edge.sx), dpp_down_inplace_f(t.sy, edge.sy);
    dpp_down_inplace_f(t.lx, edge.lx), dpp_down_inplace_f(t.ly, edge.ly);
    dpp_down_inplace(t.e, uni(edge.e));
}

// The per-step barrier of the workgroup.  Only LDS traffic (the edge cells, the pair counter) has to be complete
// before it: __syncthreads() would also wait for the forward-row stores of this step and for the forward rows the
// backward sweep loads one anti-diagonal AHEAD.  (A wavefront only ever loads rows it stored itself, so no
// global-memory ordering between wavefronts is needed.)  Measured neutral on MI355X -- the step is bound by the
// wavefront's own instruction latency -- but it is the barrier the algorithm needs.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The edge cells of the workgroup's wavefronts in LDS: [parity of the anti-diagonal][wavefront + 1][side][8 floats];
// rows 0 and NW + 1 stay dead (the frame's own ends).  side 0 = the wavefront's slot 0, side 1 = its top slot.
template <int NW>
struct Edges {
    float *base;
    __device__ __forceinline__ float *at(int par, int row, int side) const { return base + ((par * (NW + 2) + row) * 2 + side) * 8; }
    __device__ __forceinline__ Cell get(int par, int row, int side) const {
        const float4 q = *reinterpret_cast<const float4 *>(at(par, row, side));
        const float2 g = *reinterpret_cast<const float2 *>(at(par, row, side) + 4);
        return Cell{q.x, q.y, q.z, q.w, g.x, fbits(g.y)};
    }
    __device__ __forceinline__ void put(int par, int row, int side, const Cell &c) const {
        *reinterpret_cast<float4 *>(at(par, row, side)) = make_float4(c.m, c.sx, c.sy, c.lx);
        *reinterpret_cast<float2 *>(at(par, row, side) + 4) = make_float2(c.ly, bitsf(c.e));
    }
    static constexpr int floats() { return 2 * (NW + 2) * 2 * 8; }
};

// this wavefront's two edge cells of anti-diagonal `g` into LDS (lane 0 holds slot 0, lane 63 the top slot)
template <int R, int NW>
__device__ __forceinline__ void publish(const Edges<NW> &ed, int par, int wv, const Diag<R> &g) {
    if (__builtin_amdgcn_inverse_ballot_w64(1ull)) ed.put(par, wv + 1, 0, g.c[0]);
    if (__builtin_amdgcn_inverse_ballot_w64(1ull << 63)) ed.put(par, wv + 1, 1, g.c[R - 1]);
}

// band of an anti-diagonal in this wavefront's own slot numbering: [lo, lo + n) with n possibly 0
struct LocalBand {
    int lo, n;
};
template <int R>
__device__ __forceinline__ LocalBand local_band(const Ctl &ct, int sb) {
    const int jl = ct.jlo - sb;
    const int lo = max(jl, 0), hi = min(jl + ct.n, 64 * R);
    return LocalBand{hi > lo ? lo : 0, max(hi - lo, 0)};
}

template <int R>
__device__ __forceinline__ void store_row_w(char *F, const Diag<R> &C, const Ctl &ct, const Masks<R> &mk, int voff) {
    constexpr int SH = R == 1 ? 0 : (R == 2 ? 1 : 2);
    const __amdgpu_buffer_rsrc_t rs = row_rsrc<R>(F, ct.co, ct.jlo >> SH);  // lanes numbered across the whole frame
    if (__builtin_amdgcn_inverse_ballot_w64(mk.lanes)) {
        if constexpr (R == 1) {
            __builtin_amdgcn_raw_buffer_store_b64(v2i{fbits(C.c[0].m), C.c[0].e}, rs, voff, 0, 0);
        } else if constexpr (R == 2) {
            __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(C.c[0].m), C.c[0].e, fbits(C.c[1].m), C.c[1].e}, rs, voff, 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(C.c[0].m), C.c[0].e, fbits(C.c[1].m), C.c[1].e}, rs, voff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(C.c[2].m), C.c[2].e, fbits(C.c[3].m), C.c[3].e}, rs, voff + 16, 0, 0);
        }
    }
}
template <int R>
__device__ __forceinline__ void load_row_w(char *F, FRow<R> &f, const Ctl &ct, const Masks<R> &mk, int voff) {
    constexpr int SH = R == 1 ? 0 : (R == 2 ? 1 : 2);
    const __amdgpu_buffer_rsrc_t rs = row_rsrc<R>(F, ct.co, ct.jlo >> SH);
    if (__builtin_amdgcn_inverse_ballot_w64(mk.lanes)) {
        if constexpr (R == 1) {
            const v2i q = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
            f.v[0] = bitsf(q.x), f.e[0] = q.y;
        } else if constexpr (R == 2) {
            const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
            f.v[0] = bitsf(q.x), f.e[0] = q.y, f.v[1] = bitsf(q.z), f.e[1] = q.w;
        } else {
            const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
            const v4i g = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16, 0, 0);
            f.v[0] = bitsf(q.x), f.e[0] = q.y, f.v[1] = bitsf(q.z), f.e[1] = q.w;
            f.v[2] = bitsf(g.x), f.e[2] = g.y, f.v[3] = bitsf(g.z), f.e[3] = g.w;
        }
    }
}

// A wavefront is ACTIVE on an anti-diagonal whose band, widened by four slots, touches its slots.  The band's edges move
// by at most one slot per anti-diagonal and a rebase adds one more, so when the widened band does not touch the
// wavefront, neither did the bands of the two anti-diagonals before: every cell the wavefront still holds was outside
// the band when it was made, i.e. is dead already, and so are its published edges.
__device__ __forceinline__ bool band_near(const Ctl &ct, int sb, int width) {
    return ct.jlo - 4 < sb + width && ct.jlo + ct.n + 4 > sb;
}

// =====================================================================================================================
// k_em_stair<R>: Baum-Welch E-step on the register kernel (cactus_realign --outputExpectations,
// summed by cactus_expectationMaximisation at nanopore/analyses/utils.py:509-528 -- 3 trials x 100 iterations over the
// training alignments, so this pass runs hundreds of times per trained model).  Same frame-based sweep as k_dp_stair;
//   * the forward sweep stores ALL five states of every cell (the (m, e) pairs as k_dp_stair does, plus a float4
//     (sx, sy, lx, ly) per slot in a second scratch region);
//   * the backward sweep holds the forward cells of the two anti-diagonals BELOW the current one in registers (loaded
//     for every anti-diagonal with a slot shift that undoes the frame rebases in between: rows are slot-linear in
//     memory, so a shifted row is just another base address) and, after finishing a backward cell, adds the posterior probability of
//     each of the 15 transitions INTO that cell to 15 per-lane accumulators and the emitted symbols' posterior to
//     per-lane bins in LDS (no atomics in the loop);
//   * the wavefront reduces accumulators and bins at the end of the task: one fp64 atomic per count per task.
// Expected counts agree with the fp64 oracle to <= 2e-5 relative (tests/test_gpu_em.py); the summation order differs
// from k_dp_generic<EM>, so the two are not bit-identical.
// =====================================================================================================================
// lanes whose slot R*lane + r lies in [jlo, jlo + n); jlo may be negative (a band seen from a shifted frame)
template <int R>
__device__ __forceinline__ uint64_t cell_mask(int jlo, int n, int r) {
    constexpr int SH = R == 1 ? 0 : (R == 2 ? 1 : 2);
    const int lo = max(jlo - r + R - 1, 0) >> SH, hi = max(jlo + n - r + R - 1, 0) >> SH;
    return hi > lo ? (low_lanes(hi) & ~low_lanes(lo)) : 0ull;
}

// The four other forward states of a row live in four planes of the second scratch region (plane p of cell i at float
// p * stride + i, the layout k_dp_generic<EM> uses): a lane's R slots are R consecutive floats of a plane, so every
// store / load instruction moves one contiguous 256*R-byte run per wavefront -- whole cache lines.  (A float4 per slot
// wrote each line in R instalments and quadrupled the bytes that reached HBM.)
template <int R>
__device__ __forceinline__ void plane_store(__amdgpu_buffer_rsrc_t rs, int voff, const float (&v)[R]) {
    if constexpr (R == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(fbits(v[0]), rs, voff, 0, 0);
    } else if constexpr (R == 2) {
        __builtin_amdgcn_raw_buffer_store_b64(v2i{fbits(v[0]), fbits(v[1])}, rs, voff, 0, 0);
    } else {
        __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(v[0]), fbits(v[1]), fbits(v[2]), fbits(v[3])}, rs, voff, 0, 0);
    }
}
template <int R>
__device__ __forceinline__ void plane_load(__amdgpu_buffer_rsrc_t rs, int voff, float (&v)[R]) {
    if constexpr (R == 1) {
        v[0] = bitsf(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0));
    } else if constexpr (R == 2) {
        const v2i q = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0);
        v[0] = bitsf(q.x), v[1] = bitsf(q.y);
    } else {
        const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
        v[0] = bitsf(q.x), v[1] = bitsf(q.y), v[2] = bitsf(q.z), v[3] = bitsf(q.w);
    }
}

// `mk`: the lanes of THIS wavefront that hold band cells; `glane`: the lane's number across the whole frame
template <int R>
__device__ __forceinline__ void store_row_x(char *Fx, int64_t stride, const Diag<R> &C, const Ctl &ct, const Masks<R> &mk, int glane) {
    constexpr int SH = R == 1 ? 0 : (R == 2 ? 1 : 2);
    const int lane = glane;
    const int64_t first = static_cast<int64_t>(ct.co) - R * (ct.jlo >> SH);
    if (lanes_of(mk.lanes)) {
        float sx[R], sy[R], lx[R], ly[R];
#pragma unroll
        for (int r = 0; r < R; ++r) sx[r] = C.c[r].sx, sy[r] = C.c[r].sy, lx[r] = C.c[r].lx, ly[r] = C.c[r].ly;
        plane_store<R>(__builtin_amdgcn_make_buffer_rsrc(Fx + (first) * 4, 0, -1, 0x00020000), 4 * R * lane, sx);
        plane_store<R>(__builtin_amdgcn_make_buffer_rsrc(Fx + (stride + first) * 4, 0, -1, 0x00020000), 4 * R * lane, sy);
        plane_store<R>(__builtin_amdgcn_make_buffer_rsrc(Fx + (2 * stride + first) * 4, 0, -1, 0x00020000), 4 * R * lane, lx);
        plane_store<R>(__builtin_amdgcn_make_buffer_rsrc(Fx + (3 * stride + first) * 4, 0, -1, 0x00020000), 4 * R * lane, ly);
    }
}

// the forward cells of the row `ct` into G, seen from a frame in which slot j is the row's slot j + shift
// (sb: first slot of this wavefront in the frame, glane: the lane's number across the whole frame; 0 / lane in k_em_stair)
template <int R>
__device__ __forceinline__ void load_full_row(char *F, char *Fx, int64_t stride, Diag<R> &G, const Ctl &ct, int shift, int glane, int sb = 0) {
    constexpr int SH = R == 1 ? 0 : (R == 2 ? 1 : 2);
    const int lane = glane;
    uint64_t lanes = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) lanes |= cell_mask<R>(ct.jlo - shift - sb, ct.n, r);
    const int64_t first = static_cast<int64_t>(ct.co) - R * (ct.jlo >> SH) + shift;  // cell index of this frame's slot 0
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(F + first * 8, 0, -1, 0x00020000);
    if (lanes_of(lanes)) {
        float sx[R], sy[R], lx[R], ly[R];
        plane_load<R>(__builtin_amdgcn_make_buffer_rsrc(Fx + (first) * 4, 0, -1, 0x00020000), 4 * R * lane, sx);
        plane_load<R>(__builtin_amdgcn_make_buffer_rsrc(Fx + (stride + first) * 4, 0, -1, 0x00020000), 4 * R * lane, sy);
        plane_load<R>(__builtin_amdgcn_make_buffer_rsrc(Fx + (2 * stride + first) * 4, 0, -1, 0x00020000), 4 * R * lane, lx);
        plane_load<R>(__builtin_amdgcn_make_buffer_rsrc(Fx + (3 * stride + first) * 4, 0, -1, 0x00020000), 4 * R * lane, ly);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const v2i q = __builtin_amdgcn_raw_buffer_load_b64(rs, 8 * (R * lane + r), 0, 0);
            G.c[r].m = bitsf(q.x), G.c[r].e = q.y;
            G.c[r].sx = sx[r], G.c[r].sy = sy[r], G.c[r].lx = lx[r], G.c[r].ly = ly[r];
        }
    }
}

// per-lane emission bins in LDS: bin b of lane l at float b * 64 + l (byte b * 256 + 4 * l)
__device__ __forceinline__ float &bin_at(float *lbins, int byte_off) {
    return *reinterpret_cast<float *>(reinterpret_cast<char *>(lbins) + byte_off);
}

// Expected counts of the transitions into the cells of one anti-diagonal d (`io`: its backward cells; G1 / G2: the
// forward cells of d-1 / d-2 in the frame of d, Gs: those of d-1 one slot away -- above after an X-step into d, below
// after a Y-step; eX / eY: the bases consumed into each cell; jl1 / jl2: first band slot of d-1 / d-2 in this
// wavefront's slot numbering).  ODD: d is odd, i.e. the
// forward step into d was an X-step.
template <int R, bool ODD>
__device__ __forceinline__ void em_cells(const StepEnv &E, const Diag<R> &io, const Diag<R> &G1, const Diag<R> &Gs, const Diag<R> &G2, const Bases<R> &eX,
                                         const Bases<R> &eY, const Masks<R> &mk, int jl1, int n1, int jl2, int n2, int tot_e,
                                         float inv_tot, float (&acc)[15], float *lbins, int lane) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const Cell c = io.c[r];
            const uint64_t here = __ballot(c.e != E_DEAD) & mk.cell[r];
            const int ex4 = eX.b[r], ey4 = eY.b[r];
            const int lane4 = 4 * lane;
            float bM = 0.f, bXs = 0.f, bXl = 0.f, bYs = 0.f, bYl = 0.f;  // this cell's emission posteriors
            // (x-1, y-1) on d-2, same slot
            if (lanes_of(here & cell_mask<R>(jl2, n2, r))) {
                const Cell &Fm = G2.c[r];
                const int s = min(max(Fm.e + c.e - tot_e, -200), 200);
                const float em = *reinterpret_cast<const float *>(E.ltab + offsetof(DevModel, em) + 5 * ex4 + ey4);
                const float w = __builtin_ldexpf(em * c.m * inv_tot, s);
                const float t0 = Fm.m * E.tr.mm * w, t1 = Fm.sx * E.tr.sxm * w, t2 = Fm.sy * E.tr.sym * w, t3 = Fm.lx * E.tr.lxm * w,
                            t4 = Fm.ly * E.tr.lym * w;
                acc[0] += t0, acc[1] += t1, acc[2] += t2, acc[3] += t3, acc[4] += t4;
                bM = (t0 + t1) + (t2 + t3) + t4;
            }
            // (x-1, y) on d-1: same slot after an X-step into d, one slot below after a Y-step
            if (lanes_of(here & cell_mask<R>(ODD ? jl1 : jl1 + 1, n1, r))) {
                const Cell &Fl = ODD ? G1.c[r] : Gs.c[r];
                const int s = min(max(Fl.e + c.e - tot_e, -200), 200);
                const float g = __builtin_ldexpf(inv_tot, s);
                const float exs = *reinterpret_cast<const float *>(E.ltab + offsetof(DevModel, ex) + 20 + ex4);
                const float exl = *reinterpret_cast<const float *>(E.ltab + offsetof(DevModel, ex) + 60 + ex4);
                const float ws = exs * c.sx * g, wl = exl * c.lx * g;
                const float t0 = Fl.m * E.tr.msx * ws, t1 = Fl.sx * E.tr.sxsx * ws, t2 = Fl.sy * E.tr.sysx * ws;
                const float u0 = Fl.m * E.tr.mlx * wl, u1 = Fl.lx * E.tr.lxlx * wl;
                acc[5] += t0, acc[6] += t1, acc[7] += t2, acc[8] += u0, acc[9] += u1;
                bXs = (t0 + t1) + t2, bXl = u0 + u1;
            }
            // (x, y-1) on d-1: one slot above after an X-step into d, same slot after a Y-step
            if (lanes_of(here & cell_mask<R>(ODD ? jl1 - 1 : jl1, n1, r))) {
                const Cell &Fu = ODD ? Gs.c[r] : G1.c[r];
                const int s = min(max(Fu.e + c.e - tot_e, -200), 200);
                const float g = __builtin_ldexpf(inv_tot, s);
                const float eys = *reinterpret_cast<const float *>(E.ltab + offsetof(DevModel, ey) + 40 + ey4);
                const float eyl = *reinterpret_cast<const float *>(E.ltab + offsetof(DevModel, ey) + 80 + ey4);
                const float ws = eys * c.sy * g, wl = eyl * c.ly * g;
                const float t0 = Fu.m * E.tr.msy * ws, t1 = Fu.sy * E.tr.sysy * ws, t2 = Fu.sx * E.tr.sxsy * ws;
                const float u0 = Fu.m * E.tr.mly * wl, u1 = Fu.ly * E.tr.lyly * wl;
                acc[10] += t0, acc[11] += t1, acc[12] += t2, acc[13] += u0, acc[14] += u1;
                bYs = (t0 + t1) + t2, bYl = u0 + u1;
            }
            // the five bins of this cell (disjoint tables): all reads, then all writes -- one LDS round trip
            // per cell.  An N base goes to a scratch row (row EM_BINS).
            if (lanes_of(here)) {
                constexpr int TRASH = EM_BINS * 256;  // the scratch row: the last of the EM_BINS + 1 rows a wavefront has
                const bool nx = ex4 >= 16, ny = ey4 >= 16;
                const int aM = ((nx || ny) ? TRASH : ex4 * 256 + ey4 * 64) + lane4;
                const int aXs = (nx ? TRASH : 16 * 256 + ex4 * 64) + lane4, aXl = (nx ? TRASH : 20 * 256 + ex4 * 64) + lane4;
                const int aYs = (ny ? TRASH : 24 * 256 + ey4 * 64) + lane4, aYl = (ny ? TRASH : 28 * 256 + ey4 * 64) + lane4;
                const float v0 = bin_at(lbins, aM), v1 = bin_at(lbins, aXs), v2 = bin_at(lbins, aXl), v3 = bin_at(lbins, aYs),
                            v4 = bin_at(lbins, aYl);
                bin_at(lbins, aM) = v0 + bM;
                bin_at(lbins, aXs) = v1 + bXs;
                bin_at(lbins, aXl) = v2 + bXl;
                bin_at(lbins, aYs) = v3 + bYs;
                bin_at(lbins, aYl) = v4 + bYl;
            }
        }
}

// EM: the Baum-Welch E-step variant (k_em_stair's accumulation on this kernel's sweep; the trainer's own band --
// splitMatrixBiggerThanThis 300, utils.py:511 -- is 256-310 cells wide, just too wide for one wavefront).  Forward
// cells of the neighbouring anti-diagonals come from memory with a slot shift, the one-slot-away copy too, so nothing
// but the backward cells crosses wavefronts.
template <int R, int NW, bool EM = false>
__global__ void __launch_bounds__(WAVE *NW) __attribute__((amdgpu_waves_per_eu((!EM && R == 4 && NW == 8) ? 4 : 1))) k_dp_wide(KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lmodel = reinterpret_cast<float *>(smem);
    int *lmisc = reinterpret_cast<int *>(lmodel + MODEL_FLOATS);  // [0..3] totals, [4] pair counter, [5] next task
    const Edges<NW> ed{reinterpret_cast<float *>(lmisc + 8)};
    float *const lbins = ed.base + Edges<NW>::floats() + (threadIdx.x >> 6) * (EM_BINS + 1) * WAVE;  // EM: this wavefront's bins

    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = uni(static_cast<int>(threadIdx.x) >> 6);
    const int sb = wv * 64 * R;  // first slot of this wavefront
    char *const F = a.F + static_cast<int64_t>(a.slot_base + blockIdx.x) * a.slot_stride * 8;
    const int voff = 8 * R * (64 * wv + lane);
    char *const Fx = EM ? reinterpret_cast<char *>(a.Fx) + static_cast<int64_t>(blockIdx.x) * a.slot_stride * 16 : nullptr;
    int jr[R];  // slot numbers across the whole frame
#pragma unroll
    for (int r = 0; r < R; ++r) jr[r] = sb + R * lane + r;

    int t = blockIdx.x;
    while (t < a.ntasks) {
        const Task *tp = a.tasks + t;
        const int64_t x_off = uni64(tp->x_off), y_off = uni64(tp->y_off), ctl_off = uni64(tp->ctl_off),
                      pair_off = uni64(tp->pair_off);
        const int lX = uni(tp->lX), lY = uni(tp->lY), D = uni(tp->D), pair_cap = uni(tp->pair_cap),
                  flags = uni(tp->flags), model = uni(tp->model), xs = uni(tp->xs), ys = uni(tp->ys);
        cptr32 ctl = (cptr32)(a.ctl + 2 * ctl_off);
        const int rs = flags & 1, re = (flags >> 1) & 1;

        __syncthreads();
        {
            const float *gm = reinterpret_cast<const float *>(a.models + model);
            for (int i = threadIdx.x; i < MODEL_FLOATS; i += WAVE * NW) lmodel[i] = gm[i];
            // every edge cell dead
            for (int i = threadIdx.x; i < Edges<NW>::floats() / 8; i += WAVE * NW) {
                float *c = ed.base + 8 * i;
                c[0] = c[1] = c[2] = c[3] = c[4] = 0.f, c[5] = bitsf(E_DEAD), c[6] = c[7] = 0.f;
            }
            if (threadIdx.x == 0) lmisc[0] = 0, lmisc[1] = E_DEAD, lmisc[2] = 0, lmisc[3] = E_DEAD, lmisc[4] = 0;
            if constexpr (EM)
                for (int i = 0; i < EM_BINS; ++i) lbins[i * WAVE + lane] = 0.f;
        }
        __syncthreads();
        StepEnv E;
        E.mdl = reinterpret_cast<const DevModel *>(lmodel);
        E.ltab = reinterpret_cast<const char *>(lmodel);
        E.X = a.seq + x_off, E.Y = a.seq + y_off, E.lX = lX, E.lY = lY, E.lane = lane;
        {
            Trans tr = load_trans(E.mdl->T);
            if constexpr (R >= NPR_T_SGPR_MIN_R) {
                tr.mm = unif(tr.mm), tr.sxm = unif(tr.sxm), tr.sym = unif(tr.sym), tr.lxm = unif(tr.lxm), tr.lym = unif(tr.lym);
                tr.msx = unif(tr.msx), tr.sxsx = unif(tr.sxsx), tr.sysx = unif(tr.sysx);
                tr.msy = unif(tr.msy), tr.sysy = unif(tr.sysy), tr.sxsy = unif(tr.sxsy);
                tr.mlx = unif(tr.mlx), tr.lxlx = unif(tr.lxlx), tr.mly = unif(tr.mly), tr.lyly = unif(tr.lyly);
            }
            E.tr = tr;
        }
        const DevModel *mdl = E.mdl;

        // (re)build this wavefront's streams for the frame at (x0, y0); OFF = -1: forward sweep (X[x-1], Y[y-1] per
        // slot), 0: backward sweep (X[x], Y[y]).  The captured bases are what sits just outside the wavefront.
        Streams<R> S;
        auto fwd_streams = [&](int x0, int y0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                S.X.b[r] = base4(E.X, lX, x0 + jr[r] - 1);
                S.Y.b[r] = base4(E.Y, lY, y0 - jr[r] - 1);
            }
            feed_init<+1>(S.fx, E.X, lX, x0 + sb + 64 * R - 1, lane);
            feed_init<+1>(S.fy, E.Y, lY, y0 - sb, lane);
            S.xcap = uni(base4(E.X, lX, x0 + sb - 2));
            S.ycap = uni(base4(E.Y, lY, y0 - sb - 64 * R - 1));
        };
        auto bwd_streams = [&](int x0, int y0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                S.X.b[r] = base4(E.X, lX, x0 + jr[r]);
                S.Y.b[r] = base4(E.Y, lY, y0 - jr[r]);
            }
            feed_init<-1>(S.fx, E.X, lX, x0 + sb - 1, lane);
            feed_init<-1>(S.fy, E.Y, lY, y0 - sb - 64 * R, lane);
            S.xcap = uni(base4(E.X, lX, x0 + sb + 64 * R));
            S.ycap = uni(base4(E.Y, lY, y0 - sb + 1));
        };

        // =============================== forward ===============================
        Diag<R> A = dead_diag<R>(), B = dead_diag<R>();
        const Ctl c0 = read_ctl(ctl, 0);
        const int j0 = c0.jlo;
        int x0 = -j0, y0 = j0;
        bool live = band_near(c0, sb, 64 * R);  // this wavefront's registers / streams are current
        if (live) fwd_streams(x0, y0);
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (jr[r] == j0) {
                Cell c;
                c.m = mdl->start[rs * 5 + 0], c.sx = mdl->start[rs * 5 + 1], c.sy = mdl->start[rs * 5 + 2];
                c.lx = mdl->start[rs * 5 + 3], c.ly = mdl->start[rs * 5 + 4];
                normalise(c, 0);
                A.c[r] = c;
            }
        {
            const LocalBand lb = local_band<R>(c0, sb);
            store_row_w<R>(F, A, c0, band_masks<R>(lb.lo, lb.n), voff);
            if constexpr (EM) store_row_x<R>(Fx, a.slot_stride, A, c0, band_masks<R>(lb.lo, lb.n), 64 * wv + lane);
            if (live) publish<R, NW>(ed, 0, wv, A);
        }
        __syncthreads();

        // one forward step into anti-diagonal d (X-step when d is odd): `io` holds d-2 / d, `p1` holds d-1
        auto fwd = [&](int d, Diag<R> &io, Diag<R> &p1, const Ctl &ct) {
            const int par = d & 1;
            if (ct.reb) {  // uniform over the workgroup
                if (live) {
                    if (ct.reb > 0) {
                        diag_up_inplace<R>(A, ed.get(0, wv + 2, 0)), diag_up_inplace<R>(B, ed.get(1, wv + 2, 0));
                    } else {
                        diag_down_inplace<R>(A, ed.get(0, wv, 1)), diag_down_inplace<R>(B, ed.get(1, wv, 1));
                    }
                }
                x0 += ct.reb, y0 -= ct.reb;
                if (live) {
                    if (ct.reb > 0) {
                        bases_up_inplace<R>(S.X, feed_get<+1>(S.fx, E.X, E.lX, x0 + sb + 64 * R - 2, lane));
                        bases_up_inplace<R>(S.Y, S.ycap);
                    } else {
                        bases_down_inplace<R>(S.X, S.xcap);
                        bases_down_inplace<R>(S.Y, feed_get<+1>(S.fy, E.Y, E.lY, y0 - sb - 1, lane));
                    }
                }
                lds_barrier();  // everybody has read the old edges
                if (live) publish<R, NW>(ed, 0, wv, A), publish<R, NW>(ed, 1, wv, B);
                lds_barrier();
            }
            const bool act = band_near(ct, sb, 64 * R);
            if (act) {
                if (!live) {  // the band has come back into this wavefront: restart from dead cells
                    A = dead_diag<R>(), B = dead_diag<R>();
                    fwd_streams(x0, y0);
                }
                const LocalBand lb = local_band<R>(ct, sb);
                const Masks<R> mk = band_masks<R>(lb.lo, lb.n);
                if (par) {
                    S.xcap = __builtin_amdgcn_readlane(S.X.b[0], 0);
                    bases_up<R>(S.X, feed_get<+1>(S.fx, E.X, E.lX, x0 + 1 + sb + 64 * R - 2, lane));
                    const Diag<R> U = shift_up<R>(p1, ed.get(par ^ 1, wv + 2, 0));
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float em, exs, exl, eys, eyl;
                        emissions<R>(E, S.X, S.Y, r, em, exs, exl, eys, eyl);
                        Cell c = fwd_cell_dyn(norm_diag(d), E.tr, p1.c[r], io.c[r], U.c[r], em, exs, exl, eys, eyl);
                        kill_outside(c, mk.cell[r]);
                        io.c[r] = c;
                    }
                } else {
                    S.ycap = __builtin_amdgcn_readlane(S.Y.b[R - 1], 63);
                    bases_down<R>(S.Y, feed_get<+1>(S.fy, E.Y, E.lY, y0 + 1 - sb - 1, lane));
                    const Diag<R> L = shift_down<R>(p1, ed.get(par ^ 1, wv, 1));
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float em, exs, exl, eys, eyl;
                        emissions<R>(E, S.X, S.Y, r, em, exs, exl, eys, eyl);
                        Cell c = fwd_cell_dyn(norm_diag(d), E.tr, L.c[r], io.c[r], p1.c[r], em, exs, exl, eys, eyl);
                        kill_outside(c, mk.cell[r]);
                        io.c[r] = c;
                    }
                }
                store_row_w<R>(F, io, ct, mk, voff);
                if constexpr (EM) store_row_x<R>(Fx, a.slot_stride, io, ct, mk, 64 * wv + lane);
                publish<R, NW>(ed, par, wv, io);
            }
            live = act;
            if (par) x0 += 1; else y0 += 1;
            lds_barrier();
        };

        Ctl nx = c0;
        if (D >= 1) nx = read_ctl(ctl, 1);
        int d = 1;
        for (; d + 1 <= D; d += 2) {
            Ctl cur = nx;
            nx = read_ctl(ctl, d + 1);
            fwd(d, B, A, cur);
            cur = nx;
            if (d + 2 <= D) nx = read_ctl(ctl, d + 2);
            fwd(d + 1, A, B, cur);
        }
        if (d <= D) fwd(d, B, A, nx);

        // total probability at the end corner (lX, lY): slot lX - x0 of the last anti-diagonal
        {
            const int je = lX - x0;
            const bool oddD = D & 1;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (live && jr[r] == je) {
                    const Cell c = oddD ? B.c[r] : A.c[r];
                    const float raw = dot5(mdl->end + re * 5, c);
                    if (raw > 0.f) {
                        int k;
                        reinterpret_cast<float *>(lmisc)[0] = __builtin_frexpf(raw, &k);
                        lmisc[1] = c.e + k;
                    }
                }
        }
        __syncthreads();
        const float tot_m = unif(reinterpret_cast<float *>(lmisc)[0]);
        const int tot_e = uni(lmisc[1]);

        TaskOut out;
        out.tot_m = tot_m, out.tot_e = tot_e, out.btot_m = 0.f, out.btot_e = E_DEAD, out.npairs = 0;
        out.status = NPR_OK;
        const bool alive = tot_m > 0.f;
        if (!alive) out.status = NPR_ERR_ZERO_PROB;

        // =============================== backward + posteriors ===============================
        if (alive) {
            const float inv_tot = 1.0f / tot_m;
            const PairSink sink{a.px, a.py, a.pp, pair_off, pair_cap, xs, ys, a.threshold};
            __syncthreads();
            for (int i = threadIdx.x; i < Edges<NW>::floats() / 8; i += WAVE * NW) {
                float *c = ed.base + 8 * i;
                c[0] = c[1] = c[2] = c[3] = c[4] = 0.f, c[5] = bitsf(E_DEAD);
            }
            __syncthreads();
            A = dead_diag<R>(), B = dead_diag<R>();
            const bool oddD = D & 1;
            Ctl cur = read_ctl(ctl, D);
            live = band_near(cur, sb, 64 * R);
            if (live) bwd_streams(x0, y0);
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (x0 + jr[r] == lX) {
                    Cell c;
                    c.m = mdl->end[re * 5 + 0], c.sx = mdl->end[re * 5 + 1], c.sy = mdl->end[re * 5 + 2];
                    c.lx = mdl->end[re * 5 + 3], c.ly = mdl->end[re * 5 + 4];
                    normalise(c, 0);
                    if (oddD) B.c[r] = c; else A.c[r] = c;
                }
            if (live) publish<R, NW>(ed, D & 1, wv, oddD ? B : A);
            lds_barrier();
            // one backward step into anti-diagonal dd: undoes the rebase `reb` made before the forward step into dd + 1,
            // then that step (an X-step when dd is even)
            auto bwd = [&](int dd, Diag<R> &io, Diag<R> &s1, const Ctl &ct, int reb) {
                const int par = dd & 1;
                if (reb) {
                    if (live) {
                        if (reb > 0) {
                            diag_down_inplace<R>(A, ed.get(0, wv, 1)), diag_down_inplace<R>(B, ed.get(1, wv, 1));
                        } else {
                            diag_up_inplace<R>(A, ed.get(0, wv + 2, 0)), diag_up_inplace<R>(B, ed.get(1, wv + 2, 0));
                        }
                    }
                    x0 -= reb, y0 += reb;
                    if (live) {
                        if (reb > 0) {
                            bases_down_inplace<R>(S.X, feed_get<-1>(S.fx, E.X, E.lX, x0 + sb, lane));
                            bases_down_inplace<R>(S.Y, S.ycap);
                        } else {
                            bases_up_inplace<R>(S.X, S.xcap);
                            bases_up_inplace<R>(S.Y, feed_get<-1>(S.fy, E.Y, E.lY, y0 - sb - (64 * R - 1), lane));
                        }
                    }
                    lds_barrier();
                    if (live) publish<R, NW>(ed, 0, wv, A), publish<R, NW>(ed, 1, wv, B);
                    lds_barrier();
                }
                const bool act = band_near(ct, sb, 64 * R);
                if (par) y0 -= 1; else x0 -= 1;  // the frame of dd
                if (act) {
                    if (!live) {
                        A = dead_diag<R>(), B = dead_diag<R>();
                        bwd_streams(x0 + (par ? 0 : 1), y0 + (par ? 1 : 0));  // the frame before this step is undone
                    }
                    const LocalBand lb = local_band<R>(ct, sb);
                    const Masks<R> mk = band_masks<R>(lb.lo, lb.n);
                    if (!par) {  // undo the X-step into dd + 1
                        S.xcap = __builtin_amdgcn_readlane(S.X.b[R - 1], 63);
                        bases_down<R>(S.X, feed_get<-1>(S.fx, E.X, E.lX, x0 + sb, lane));
                        const Diag<R> Ys = shift_down<R>(s1, ed.get(par ^ 1, wv, 1));
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            float em, exs, exl, eys, eyl;
                            emissions<R>(E, S.X, S.Y, r, em, exs, exl, eys, eyl);
                            Cell c = bwd_cell_dyn(norm_diag(dd), E.tr, io.c[r], s1.c[r], Ys.c[r], em, exs, exl, eys, eyl);
                            kill_outside(c, mk.cell[r]);
                            io.c[r] = c;
                        }
                    } else {  // undo the Y-step into dd + 1
                        S.ycap = __builtin_amdgcn_readlane(S.Y.b[0], 0);
                        bases_up<R>(S.Y, feed_get<-1>(S.fy, E.Y, E.lY, y0 - sb - (64 * R - 1), lane));
                        const Diag<R> Xs = shift_up<R>(s1, ed.get(par ^ 1, wv + 2, 0));
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            float em, exs, exl, eys, eyl;
                            emissions<R>(E, S.X, S.Y, r, em, exs, exl, eys, eyl);
                            Cell c = bwd_cell_dyn(norm_diag(dd), E.tr, io.c[r], Xs.c[r], s1.c[r], em, exs, exl, eys, eyl);
                            kill_outside(c, mk.cell[r]);
                            io.c[r] = c;
                        }
                    }
                    publish<R, NW>(ed, par, wv, io);
                }
                live = act;
            };

            if constexpr (EM) {
                // ---- expected counts (see k_em_stair); q0, q1, q2: control words of d, d-1, d-2 ----
                float acc[15];
#pragma unroll
                for (int i = 0; i < 15; ++i) acc[i] = 0.f;
                Diag<R> G1 = dead_diag<R>(), Gs = dead_diag<R>(), G2 = dead_diag<R>();
                const Ctl none{0u, 0, 0, 0};
                const int glane = 64 * wv + lane;
                // forward cells for the counts of anti-diagonal d, in ITS frame: d-1 (and one slot away), d-2
                auto em_load = [&](int d, const Ctl &q0, const Ctl &q1, const Ctl &q2) {
                    if (!band_near(q0, sb, 64 * R)) return;
                    if (d >= 1) {
                        load_full_row<R>(F, Fx, a.slot_stride, G1, q1, q0.reb, glane, sb);
                        load_full_row<R>(F, Fx, a.slot_stride, Gs, q1, q0.reb + ((d & 1) ? 1 : -1), glane, sb);
                    }
                    if (d >= 2) load_full_row<R>(F, Fx, a.slot_stride, G2, q2, q1.reb + q0.reb, glane, sb);
                };
                auto em_acc = [&](int d, const Ctl &q0, const Ctl &q1, const Ctl &q2) {
                    if (!live) return;
                    const LocalBand lb = local_band<R>(q0, sb);
                    const Masks<R> mk = band_masks<R>(lb.lo, lb.n);
                    const int jl1 = q1.jlo - q0.reb - sb, jl2 = q2.jlo - q1.reb - q0.reb - sb;
                    Bases<R> eX, eY;  // X[x-1] = slot j-1 of the X stream, Y[y-1] = slot j+1 of the Y stream
                    const int injx = feed_peek<-1>(S.fx, x0 + sb - 1), injy = feed_peek<-1>(S.fy, y0 - sb - 64 * R);
#pragma unroll
                    for (int r = 1; r < R; ++r) eX.b[r] = S.X.b[r - 1];
                    eX.b[0] = dpp_from_below(S.X.b[R - 1], injx);
#pragma unroll
                    for (int r = 0; r + 1 < R; ++r) eY.b[r] = S.Y.b[r + 1];
                    eY.b[R - 1] = dpp_from_above(S.Y.b[0], injy);
                    if (d & 1) {
                        em_cells<R, true>(E, B, G1, Gs, G2, eX, eY, mk, jl1, q1.n, jl2, q2.n, tot_e, inv_tot, acc, lbins, lane);
                    } else {
                        em_cells<R, false>(E, A, G1, Gs, G2, eX, eY, mk, jl1, q1.n, jl2, q2.n, tot_e, inv_tot, acc, lbins, lane);
                    }
                };
                Ctl q0 = cur, q1 = D >= 1 ? read_ctl(ctl, D - 1) : none, q2 = D >= 2 ? read_ctl(ctl, D - 2) : none;
                em_load(D, q0, q1, q2);
                for (int d = D; d >= 1; --d) {
                    em_acc(d, q0, q1, q2);
                    const int reb = q0.reb;
                    q0 = q1, q1 = q2, q2 = d >= 3 ? read_ctl(ctl, d - 3) : none;
                    em_load(d - 1, q0, q1, q2);  // before the step: it hides the latency
                    if ((d - 1) & 1) {
                        bwd(d - 1, B, A, q0, reb);
                    } else {
                        bwd(d - 1, A, B, q0, reb);
                    }
                    lds_barrier();
                }
                // this wavefront's bins and accumulators -> the model's global counts
                __syncthreads();
                if (lane < EM_BINS) {
                    double sum = 0.0;
                    for (int q = 0; q < WAVE; ++q) sum += static_cast<double>(lbins[lane * WAVE + q]);
                    atomicAdd(a.em_E + model * EM_BINS + lane, sum);
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 15; ++i) lbins[i * WAVE + lane] = acc[i];
                __syncthreads();
                if (lane < 15) {
                    double sum = 0.0;
                    for (int q = 0; q < WAVE; ++q) sum += static_cast<double>(lbins[lane * WAVE + q]);
                    const int map[15] = {0, 5, 10, 15, 20, 1, 6, 11, 3, 18, 2, 12, 7, 4, 24};  // accumulator order -> T[from*5+to]
                    atomicAdd(a.em_T + model * 25 + map[lane], sum);
                }
                __syncthreads();
            } else {
            FRow<R> fa, fb;  // forward rows of the even / odd anti-diagonals, loaded one ahead
#pragma unroll
            for (int r = 0; r < R; ++r) fa.v[r] = fb.v[r] = 0.f, fa.e[r] = fb.e[r] = E_DEAD;
            // (a wavefront the band does not touch leaves at once: most wavefronts of a wide frame, most of the time)
            auto load = [&](FRow<R> &f, const Ctl &ct) {
                if (ct.jlo + ct.n <= sb || ct.jlo >= sb + 64 * R) return;
                const LocalBand lb = local_band<R>(ct, sb);
                load_row_w<R>(F, f, ct, band_masks<R>(lb.lo, lb.n), voff);
            };
            // posteriors of anti-diagonal dd (frame at x0, y0), slots claimed from the workgroup's LDS counter
            auto emit = [&](const Diag<R> &Bd, const FRow<R> &f, int dd, const Ctl &ct) {
                if (ct.jlo + ct.n <= sb || ct.jlo >= sb + 64 * R) return;
                const LocalBand lb = local_band<R>(ct, sb);
                const Masks<R> mk = band_masks<R>(lb.lo, lb.n);
                float p[R];
                uint64_t hit[R];
                int total = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    p[r] = posterior(f.v[r], f.e[r], Bd.c[r].m, Bd.c[r].e, tot_e, inv_tot);
                    hit[r] = __ballot(p[r] >= sink.threshold) & mk.cell[r];
                    total += __popcll(hit[r]);
                }
                if (dd >= 2 && total) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(lmisc + 4, total);
                    base = uni(base);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if (hit[r]) {
                            const int before = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(hit[r] >> 32),
                                                                         __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(hit[r]), 0));
                            const int slot = base + before;
                            if (__builtin_amdgcn_inverse_ballot_w64(hit[r]) && slot < sink.cap) {
                                sink.px[sink.off + slot] = x0 + jr[r] - 1 + sink.xs;
                                sink.py[sink.off + slot] = y0 - jr[r] - 1 + sink.ys;
                                sink.pp[sink.off + slot] = p[r];
                            }
                            base += __popcll(hit[r]);
                        }
                    }
                }
            };
            Ctl nxt = cur;
            if (oddD) {
                load(fb, cur);
                nxt = read_ctl(ctl, D - 1);
                load(fa, nxt);
                emit(B, fb, D, cur);
            } else {
                load(fa, cur);
                if (D >= 1) {
                    nxt = read_ctl(ctl, D - 1);
                    load(fb, nxt);
                }
                emit(A, fa, D, cur);
            }
            
            int d2 = D - 1;
            if (oddD) {
                const int reb = cur.reb;
                cur = nxt;
                if (d2 >= 1) {
                    nxt = read_ctl(ctl, d2 - 1);
                    load(fb, nxt);
                }
                bwd(d2, A, B, cur, reb);
                emit(A, fa, d2, cur);
                lds_barrier();
                d2 -= 1;
            }
            for (; d2 >= 1; d2 -= 2) {
                int reb = cur.reb;
                cur = nxt;
                nxt = read_ctl(ctl, d2 - 1);
                load(fa, nxt);
                bwd(d2, B, A, cur, reb);
                emit(B, fb, d2, cur);
                lds_barrier();
                reb = cur.reb;
                cur = nxt;
                if (d2 >= 2) {
                    nxt = read_ctl(ctl, d2 - 2);
                    load(fb, nxt);
                }
                bwd(d2 - 1, A, B, cur, reb);
                emit(A, fa, d2 - 1, cur);
                lds_barrier();
            }
            }
            // total from the backward side: the lattice point (0, 0) is slot j0 of anti-diagonal 0
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (live && jr[r] == j0) {
                    const Cell cz = A.c[r];
                    const float raw = dot5(mdl->start + rs * 5, cz);
                    if (raw > 0.f) {
                        int k;
                        reinterpret_cast<float *>(lmisc)[2] = __builtin_frexpf(raw, &k);
                        lmisc[3] = cz.e + k;
                    }
                }
            __syncthreads();
            out.btot_m = unif(reinterpret_cast<float *>(lmisc)[2]);
            out.btot_e = uni(lmisc[3]);
        }
        if (threadIdx.x == 0) {
            const int cnt = lmisc[4];
            out.npairs = cnt;
            if (cnt > pair_cap) out.status = NPR_ERR_CAPACITY;
            a.outs[t] = out;
            lmisc[5] = atomicAdd(a.queue, 1);
        }
        __syncthreads();
        t = uni(lmisc[5]) + static_cast<int>(gridDim.x);
    }
}


template <int R>
__global__ void __launch_bounds__(WAVE) k_em_stair(KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lmodel = reinterpret_cast<float *>(smem);
    int *lmisc = reinterpret_cast<int *>(lmodel + MODEL_FLOATS);  // 8 ints: cell hand-off
    float *lbins = reinterpret_cast<float *>(lmisc + 8);          // EM_BINS + 1 rows of 64 lanes (the last: scratch row for N bases)

    const int lane = threadIdx.x;
    char *const F = a.F + static_cast<int64_t>(a.slot_base + blockIdx.x) * a.slot_stride * 8;
    char *const Fx = reinterpret_cast<char *>(a.Fx) + static_cast<int64_t>(blockIdx.x) * a.slot_stride * 16;
    const int voff = 8 * R * lane;
    int jr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) jr[r] = R * lane + r;

    int t = blockIdx.x;
    while (t < a.ntasks) {
        const Task *tp = a.tasks + t;
        const int64_t x_off = uni64(tp->x_off), y_off = uni64(tp->y_off), ctl_off = uni64(tp->ctl_off);
        const int lX = uni(tp->lX), lY = uni(tp->lY), D = uni(tp->D), flags = uni(tp->flags), model = uni(tp->model);
        cptr32 ctl = (cptr32)(a.ctl + 2 * ctl_off);
        const int rs = flags & 1, re = (flags >> 1) & 1;

        __syncthreads();
        {
            const float *gm = reinterpret_cast<const float *>(a.models + model);
            for (int i = lane; i < MODEL_FLOATS; i += WAVE) lmodel[i] = gm[i];
            for (int i = 0; i < EM_BINS; ++i) lbins[i * WAVE + lane] = 0.f;
        }
        __syncthreads();
        StepEnv E;
        E.mdl = reinterpret_cast<const DevModel *>(lmodel);
        E.ltab = reinterpret_cast<const char *>(lmodel);
        E.X = a.seq + x_off, E.Y = a.seq + y_off, E.lX = lX, E.lY = lY, E.lane = lane;
        {
            E.tr = load_trans(E.mdl->T);  // in VGPRs: every count multiplies by one, and the scalar file is short here
        }
        const DevModel *mdl = E.mdl;

        // =============================== forward: as k_dp_stair, all five states stored ===============================
        Diag<R> A = dead_diag<R>(), B = dead_diag<R>();
        Streams<R> S;
        const Ctl c0 = read_ctl_one<R>(ctl, 0);
        const int j0 = c0.jlo;
        int x0 = -j0, y0 = j0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            S.X.b[r] = base4(E.X, lX, x0 + jr[r] - 1);
            S.Y.b[r] = base4(E.Y, lY, y0 - jr[r] - 1);
        }
        S.xcap = S.ycap = 16;
        feed_init<+1>(S.fx, E.X, lX, x0 + 64 * R - 1, lane);
        feed_init<+1>(S.fy, E.Y, lY, y0, lane);
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (jr[r] == j0) {
                Cell c;
                c.m = mdl->start[rs * 5 + 0], c.sx = mdl->start[rs * 5 + 1], c.sy = mdl->start[rs * 5 + 2];
                c.lx = mdl->start[rs * 5 + 3], c.ly = mdl->start[rs * 5 + 4];
                normalise(c, 0);
                A.c[r] = c;
            }
        store_row<R>(F, A, c0, voff), store_row_x<R>(Fx, a.slot_stride, A, c0, band_masks<R>(c0.jlo, c0.n), lane);
        for (int d = 1; d <= D; ++d) {
            const Ctl ct = read_ctl_one<R>(ctl, d);
            if (ct.reb) fwd_rebase<R>(E, ct.reb, A, B, S, x0, y0);
            if (d & 1) {
                fwd_x_step<R>(d, E, B, A, S, x0, ct);
                store_row<R>(F, B, ct, voff), store_row_x<R>(Fx, a.slot_stride, B, ct, band_masks<R>(ct.jlo, ct.n), lane);
            } else {
                fwd_y_step<R>(d, E, A, B, S, y0, ct);
                store_row<R>(F, A, ct, voff), store_row_x<R>(Fx, a.slot_stride, A, ct, band_masks<R>(ct.jlo, ct.n), lane);
            }
        }
        {
            const int je = lX - x0;
            const bool oddD = D & 1;
            float tm = 0.f;
            int te = E_DEAD;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (jr[r] == je) {
                    const Cell c = oddD ? B.c[r] : A.c[r];
                    const float raw = dot5(mdl->end + re * 5, c);
                    if (raw > 0.f) {
                        int k;
                        tm = __builtin_frexpf(raw, &k);
                        te = c.e + k;
                    }
                    reinterpret_cast<float *>(lmisc)[0] = tm;
                    lmisc[1] = te;
                }
        }
        __syncthreads();
        const float tot_m = unif(reinterpret_cast<float *>(lmisc)[0]);
        const int tot_e = uni(lmisc[1]);
        __syncthreads();

        TaskOut out;
        out.tot_m = tot_m, out.tot_e = tot_e, out.btot_m = 0.f, out.btot_e = E_DEAD, out.npairs = 0;
        out.status = NPR_OK;
        const bool alive = tot_m > 0.f;
        if (!alive) out.status = NPR_ERR_ZERO_PROB;

        // =============================== backward + expected counts ===============================
        if (alive) {
            const float inv_tot = 1.0f / tot_m;
            float acc[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) acc[i] = 0.f;
            A = dead_diag<R>(), B = dead_diag<R>();
            // q0..q3: control words of the anti-diagonals d, d-1, d-2, d-3 (n = 0 below the first one)
            const Ctl none{0u, 0, 0, 0};
            Ctl q0 = read_ctl_one<R>(ctl, D), q1 = D >= 1 ? read_ctl_one<R>(ctl, D - 1) : none, q2 = D >= 2 ? read_ctl_one<R>(ctl, D - 2) : none,
                q3 = D >= 3 ? read_ctl_one<R>(ctl, D - 3) : none;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                S.X.b[r] = base4(E.X, lX, x0 + jr[r]);
                S.Y.b[r] = base4(E.Y, lY, y0 - jr[r]);
                if (x0 + jr[r] == lX) {
                    Cell c;
                    c.m = mdl->end[re * 5 + 0], c.sx = mdl->end[re * 5 + 1], c.sy = mdl->end[re * 5 + 2];
                    c.lx = mdl->end[re * 5 + 3], c.ly = mdl->end[re * 5 + 4];
                    normalise(c, 0);
                    if (D & 1) B.c[r] = c; else A.c[r] = c;
                }
            }
            S.xcap = S.ycap = 16;
            feed_init<-1>(S.fx, E.X, lX, x0 - 1, lane);
            feed_init<-1>(S.fy, E.Y, lY, y0 - 64 * R, lane);
            // forward cells of d-1 and d-2 in the frame of d.  Both are loaded afresh for every anti-diagonal (right after
            // their last use, so the backward step hides the latency): a row loaded once and then carried along through
            // the rebases would lose the cells that lie outside one frame but inside the next.
            Diag<R> G1 = dead_diag<R>(), G2 = dead_diag<R>();
            if (D >= 1) load_full_row<R>(F, Fx, a.slot_stride, G1, q1, q0.reb, lane);
            if (D >= 2) load_full_row<R>(F, Fx, a.slot_stride, G2, q2, q1.reb + q0.reb, lane);

            for (int d = D; d >= 1; --d) {
                // ---- expected counts of the transitions into the cells of d ----
                const Masks<R> mk = band_masks<R>(q0.jlo, q0.n);
                const int jl1 = q1.jlo - q0.reb, jl2 = q2.jlo - q1.reb - q0.reb;
                // bases consumed INTO (x, y): X[x-1] = slot j-1 of the X stream, Y[y-1] = slot j+1 of the Y stream
                Bases<R> eX, eY;
                {
                    const int injx = feed_peek<-1>(S.fx, x0 - 1), injy = feed_peek<-1>(S.fy, y0 - 64 * R);
#pragma unroll
                    for (int r = 1; r < R; ++r) eX.b[r] = S.X.b[r - 1];
                    eX.b[0] = dpp_from_below(S.X.b[R - 1], injx);
#pragma unroll
                    for (int r = 0; r + 1 < R; ++r) eY.b[r] = S.Y.b[r + 1];
                    eY.b[R - 1] = dpp_from_above(S.Y.b[0], injy);
                }
                if (d & 1) {
                    em_cells<R, true>(E, B, G1, shift_up<R>(G1), G2, eX, eY, mk, jl1, q1.n, jl2, q2.n, tot_e, inv_tot, acc, lbins, lane);
                } else {
                    em_cells<R, false>(E, A, G1, shift_down<R>(G1), G2, eX, eY, mk, jl1, q1.n, jl2, q2.n, tot_e, inv_tot, acc, lbins, lane);
                }
                // ---- on to anti-diagonal d-1: undo the rebase made before the forward step into d, then that step ----
                // forward cells for the next anti-diagonal, d-1: those of d-2 and d-3 in ITS frame
                if (d >= 2) load_full_row<R>(F, Fx, a.slot_stride, G1, q2, q1.reb, lane);
                if (d >= 3) load_full_row<R>(F, Fx, a.slot_stride, G2, q3, q2.reb + q1.reb, lane);
                if (q0.reb) bwd_rebase<R>(E, q0.reb, A, B, S, x0, y0);
                if ((d - 1) & 1) {
                    bwd_y_step<R>(d - 1, E, B, A, S, y0, q1);
                } else {
                    bwd_x_step<R>(d - 1, E, A, B, S, x0, q1);
                }
                q0 = q1, q1 = q2, q2 = q3, q3 = d >= 4 ? read_ctl_one<R>(ctl, d - 4) : none;
            }
            // total from the backward side
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (jr[r] == j0) {
                    const Cell cz = A.c[r];
                    const float raw = dot5(mdl->start + rs * 5, cz);
                    float bm = 0.f;
                    int be = E_DEAD;
                    if (raw > 0.f) {
                        int k;
                        bm = __builtin_frexpf(raw, &k);
                        be = cz.e + k;
                    }
                    reinterpret_cast<float *>(lmisc)[2] = bm;
                    lmisc[3] = be;
                }
            __syncthreads();
            out.btot_m = unif(reinterpret_cast<float *>(lmisc)[2]);
            out.btot_e = uni(lmisc[3]);
            // emission bins: one lane per bin sums its 64 columns; then the 15 transition accumulators go through the
            // same rows
            __syncthreads();
            if (lane < EM_BINS) {
                double sum = 0.0;
                for (int q = 0; q < WAVE; ++q) sum += static_cast<double>(lbins[lane * WAVE + q]);
                atomicAdd(a.em_E + model * EM_BINS + lane, sum);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 15; ++i) lbins[i * WAVE + lane] = acc[i];
            __syncthreads();
            if (lane < 15) {
                double sum = 0.0;
                for (int q = 0; q < WAVE; ++q) sum += static_cast<double>(lbins[lane * WAVE + q]);
                const int map[15] = {0, 5, 10, 15, 20, 1, 6, 11, 3, 18, 2, 12, 7, 4, 24};  // accumulator order -> T[from*5+to]
                atomicAdd(a.em_T + model * 25 + map[lane], sum);
            }
            __syncthreads();
        }
        if (lane == 0) a.outs[t] = out;
        int nt = 0;
        if (lane == 0) nt = atomicAdd(a.queue, 1);
        t = uni(nt) + static_cast<int>(gridDim.x);
    }
}

}  // namespace

size_t stair_lds_bytes() { return sizeof(float) * (MODEL_FLOATS + 8); }

size_t wide_lds_bytes(int nw) { return sizeof(float) * (MODEL_FLOATS + 8 + 2 * (nw + 2) * 2 * 8); }

// (R, NW) pairs built: 2x4 = 512 slots, 2x8 = 1024, 4x8 = 2048, 4x12 = 3072.  What bounds a step is ONE wavefront's
// latency (LDS edge read -> DPP -> ~50 dependent VALU ops -> publish -> barrier), so a wavefront should carry enough
// slots to amortise it (one slot per lane: 6e10 cells/s on constant 400-800-cell bands, two: 1.1-1.3e11), and a CU
// should hold more than one task to fill the time the others wait: 2048 slots as 2x16 (one workgroup per CU) gives
// 6.1e10 cells/s on the reference's anchor diamonds, as 4x8 squeezed into 128 VGPRs (68 spilled registers, but two
// workgroups per CU) 6.5e10.
int launch_wide(const KernelArgs &a, int R, int NW, int grid, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = wide_lds_bytes(NW);
    if (R == 2 && NW == 4)
        hipLaunchKernelGGL((k_dp_wide<2, 4>), dim3(grid), dim3(WAVE * 4), lds, s, a);
    else if (R == 2 && NW == 8)
        hipLaunchKernelGGL((k_dp_wide<2, 8>), dim3(grid), dim3(WAVE * 8), lds, s, a);
    else if (R == 4 && NW == 8)
        hipLaunchKernelGGL((k_dp_wide<4, 8>), dim3(grid), dim3(WAVE * 8), lds, s, a);
    else if (R == 4 && NW == 12)
        hipLaunchKernelGGL((k_dp_wide<4, 12>), dim3(grid), dim3(WAVE * 12), lds, s, a);
    else
        return static_cast<int>(hipErrorInvalidValue);
    return static_cast<int>(hipGetLastError());
}

size_t em_wide_lds_bytes(int nw) { return wide_lds_bytes(nw) + sizeof(float) * static_cast<size_t>(nw) * (EM_BINS + 1) * WAVE; }

int launch_em_wide(const KernelArgs &a, int R, int NW, int grid, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = em_wide_lds_bytes(NW);
    if (R == 2 && NW == 4)
        hipLaunchKernelGGL((k_dp_wide<2, 4, true>), dim3(grid), dim3(WAVE * 4), lds, s, a);
    else if (R == 2 && NW == 8)
        hipLaunchKernelGGL((k_dp_wide<2, 8, true>), dim3(grid), dim3(WAVE * 8), lds, s, a);
    else
        return static_cast<int>(hipErrorInvalidValue);
    return static_cast<int>(hipGetLastError());
}

size_t em_stair_lds_bytes() { return sizeof(float) * (MODEL_FLOATS + 8 + (EM_BINS + 1) * WAVE); }

int launch_em_stair(const KernelArgs &a, int R, int grid, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = em_stair_lds_bytes();
    if (R == 1)
        hipLaunchKernelGGL(k_em_stair<1>, dim3(grid), dim3(WAVE), lds, s, a);
    else if (R == 2)
        hipLaunchKernelGGL(k_em_stair<2>, dim3(grid), dim3(WAVE), lds, s, a);
    else if (R == 4)
        hipLaunchKernelGGL(k_em_stair<4>, dim3(grid), dim3(WAVE), lds, s, a);
    else
        return static_cast<int>(hipErrorInvalidValue);
    return static_cast<int>(hipGetLastError());
}

int launch_stair(const KernelArgs &a, int R, int grid, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = stair_lds_bytes();
    if (R == 1)
        hipLaunchKernelGGL(k_dp_stair<1>, dim3(grid), dim3(WAVE), lds, s, a);
    else if (R == 2)
        hipLaunchKernelGGL(k_dp_stair<2>, dim3(grid), dim3(WAVE), lds, s, a);
    else if (R == 4)
        hipLaunchKernelGGL(k_dp_stair<4>, dim3(grid), dim3(WAVE), lds, s, a);
    else
        return static_cast<int>(hipErrorInvalidValue);
    return static_cast<int>(hipGetLastError());
}

}  // namespace npr
