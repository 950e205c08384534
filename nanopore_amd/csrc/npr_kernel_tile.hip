// npr_kernel_tile.hip -- k_dp_tile<R>: the register-resident DP kernel for WIDE bands.
//
// Same recurrences and the same per-cell arithmetic (npr_cell.h) as k_dp_stair / k_dp_generic -- cactus_realign's banded
// five-state forward / backward / posterior pass, SURVEY.md 8a rows a5.3-a5.5, reference call site
// nanopore/analyses/utils.py:587 -- for the band the reference's own parameters give (anchors +- diagonalExpansion 10,
// 14 trimmed columns, splitMatrixBiggerThanThis 3000): a chain of unanchored rectangles hundreds to 3000 cells across,
// joined by 21-cell stripes.  97 % of its cells lie on anti-diagonals wider than one wavefront can hold.
//
// A frame that follows the anti-diagonal (k_dp_stair, k_dp_wide) needs the neighbour on BOTH sides, so a band spread
// over several wavefronts has to synchronise all of them after every anti-diagonal.  In lattice coordinates the
// dependencies of a cell, (x-1, y), (x, y-1), (x-1, y-1), point one way only.  So here the lattice COLUMNS of a task
// are cut into stripes of 64*R columns and one wavefront sweeps one stripe:
//   * slot j of the wavefront is lattice column X + j for the whole life of the stripe; on anti-diagonal d it holds the
//     cell (X + j, d - X - j).  (x, y-1) is the same slot on d-1, (x-1, y) the slot below on d-1, (x-1, y-1) the slot
//     below on d-2: one DPP shift per anti-diagonal (the shifted copy of d-1 is kept for the next step, where it is the
//     shifted d-2), no frame rebases, no reference stream -- a slot's reference base never changes; the read streams
//     through the wavefront one slot per step;
//   * the only thing a stripe needs from outside is the last column of the stripe to its left (forward) or the first
//     column of the stripe to its right (backward): one cell per anti-diagonal.  The producer writes it to the task's
//     scratch, the consumer stages 16 of them at a time into LDS; a per-wavefront progress word in LDS says how far a
//     stripe has got.  Wavefronts of a workgroup take the stripes of ONE task round-robin and run as a pipeline, each a
//     stripe width behind its left neighbour -- no barrier inside the sweep;
//   * forward match values stream to HBM one fixed-stride row per anti-diagonal of a stripe and stream back one row
//     ahead of use (every stripe is 64*R columns wide -- the last one may reach past lX --, so the right neighbour's
//     column always enters at lane 63 and a row is read by the lanes that wrote it);
//   * which lanes of a row hold band cells is read from a table the planner made (one packed word per row, k_plan_rowmask),
//     rows and neighbour cells are addressed by one buffer descriptor per stripe plus a vector offset: next to nothing
//     per step on the scalar unit, which the CU's four SIMDs share.
// Any band shape and width goes: what is outside the band is masked per anti-diagonal from the band arrays.
// Bit-identical to the other kernels and to the fp32 mirror (same cell arithmetic, order of evaluation is irrelevant to
// a cell's value).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "npr_cell.h"
#include "npr_device.h"
#include "npr_frame.h"

namespace npr {

namespace {

constexpr int TILE_MAX_NW = 8;  // wavefronts per workgroup (launch bound)
constexpr int TILE_BLOCK = 16;  // neighbour cells staged / published at a time
constexpr int EDGE_FLOATS = 8;  // one neighbour cell in memory: m, sx, sy, lx, ly, e, -, -
#ifndef NPR_EDGE_ST_AUX
#define NPR_EDGE_ST_AUX 0
#endif
#ifndef NPR_EDGE_LD_AUX
#define NPR_EDGE_LD_AUX 16
#endif

typedef const __attribute__((address_space(4))) int32_t *cptr_i32;

struct UStripe {
    int X, K, df, dl;
    uint32_t row0;
};
__device__ __forceinline__ UStripe load_stripe(const Stripe *tab, int s) {
    cptr_i32 p = (cptr_i32)(tab + s);
    return UStripe{p[0], p[1], p[2], p[3], static_cast<uint32_t>(p[4])};
}

// lane masks of a row from its packed word (npr_sched.h tile_row_word): mask_r = (~0 << lo_r) & (~0 >> sh_r), six-bit
// fields -- the 64-bit shifts take six bits of their count, so the fields need no masking
__device__ __forceinline__ Masks<2> row_masks(uint32_t w) {
    Masks<2> m;
    m.cell[0] = (~0ull << (w & 63u)) & (~0ull >> ((w >> 6) & 63u));
    m.cell[1] = (~0ull << ((w >> 12) & 63u)) & (~0ull >> ((w >> 18) & 63u));
    m.lanes = m.cell[0] | m.cell[1];
    m.l0 = 0;
    return m;
}

// progress words: volatile LDS accesses (the pointer has to say LDS, or the compiler emits flat loads)
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ int lds_peek(const int *p) { return *(const volatile lds_int *)(p); }
__device__ __forceinline__ void lds_poke(int *p, int v) { *(volatile lds_int *)(p) = v; }

// One row per anti-diagonal of a stripe: 64*R cells of 8 bytes, lane l at 8*R*l.  The descriptor is the stripe's (its
// first row); `vo` = this lane's byte offset in the row + 8 * 64 * R * (row - first row), in a VGPR: nothing per row on the
// scalar unit, and no scalar offset operand (npr_frame.h store_row explains why).
template <int R>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stripe_rsrc(char *base, uint32_t row0, int row_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base + static_cast<int64_t>(row0) * row_bytes, 0, -1, 0x00020000);
}
template <int R>
__device__ __forceinline__ void tile_store_row(__amdgpu_buffer_rsrc_t rs, int vo, const Diag<R> &C, const Masks<R> &mk) {
    static_assert(R == 2, "k_dp_tile: two slots per lane");
    if (__builtin_amdgcn_inverse_ballot_w64(mk.lanes))
        __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(C.c[0].m), C.c[0].e, fbits(C.c[1].m), C.c[1].e}, rs, vo, 0, 0);
}
template <int R>
__device__ __forceinline__ void tile_load_row(__amdgpu_buffer_rsrc_t rs, int vo, FRow<R> &f, const Masks<R> &mk) {
    if (__builtin_amdgcn_inverse_ballot_w64(mk.lanes)) {
        const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0);
        f.v[0] = bitsf(q.x), f.e[0] = q.y, f.v[1] = bitsf(q.z), f.e[1] = q.w;
    }
}

// the neighbour cell of a row: 32 bytes at (stripe's first) + 32 * (row - first row), written by the one lane that holds it
__device__ __forceinline__ void edge_store(__amdgpu_buffer_rsrc_t rs, int k, const Cell &c, uint64_t lane_mask) {
    if (__builtin_amdgcn_inverse_ballot_w64(lane_mask)) {
        const int vo = 4 * EDGE_FLOATS * k;
        __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(c.m), fbits(c.sx), fbits(c.sy), fbits(c.lx)}, rs, vo, 0, NPR_EDGE_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b64(v2i{fbits(c.ly), c.e}, rs, vo + 16, 0, NPR_EDGE_ST_AUX);
    }
}
// `cnt` neighbour cells starting at row `row` into this wavefront's LDS staging (lane l takes cell l).  The loads bypass
// the vector L1 (sc1: agent scope): the producer is another wavefront of this workgroup and its stores went through that cache.
__device__ __forceinline__ void edge_stage(char *Eb, uint32_t row, int cnt, float *stage, int lane) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Eb + static_cast<int64_t>(row) * (4 * EDGE_FLOATS), 0, -1, 0x00020000);
    if (__builtin_amdgcn_inverse_ballot_w64(low_lanes(cnt))) {
        const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rs, 32 * lane, 0, NPR_EDGE_LD_AUX);
        const v2i g = __builtin_amdgcn_raw_buffer_load_b64(rs, 32 * lane + 16, 0, NPR_EDGE_LD_AUX);
        *reinterpret_cast<v4i *>(stage + EDGE_FLOATS * lane) = q;
        *reinterpret_cast<v2i *>(stage + EDGE_FLOATS * lane + 4) = g;
    }
}
__device__ __forceinline__ Cell edge_get(const float *stage, int k) {
    const float4 q = *reinterpret_cast<const float4 *>(stage + EDGE_FLOATS * k);
    const float2 g = *reinterpret_cast<const float2 *>(stage + EDGE_FLOATS * k + 4);
    return Cell{q.x, q.y, q.z, q.w, g.x, fbits(g.y)};
}

__device__ __forceinline__ Cell dpp_cell_from_below(const Cell &v, const Cell &edge) {
    Cell o;
    o.m = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.m), fbits(v.m), 0x138, 0xf, 0xf, false));
    o.sx = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.sx), fbits(v.sx), 0x138, 0xf, 0xf, false));
    o.sy = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.sy), fbits(v.sy), 0x138, 0xf, 0xf, false));
    o.lx = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.lx), fbits(v.lx), 0x138, 0xf, 0xf, false));
    o.ly = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.ly), fbits(v.ly), 0x138, 0xf, 0xf, false));
    o.e = __builtin_amdgcn_update_dpp(edge.e, v.e, 0x138, 0xf, 0xf, false);
    return o;
}
__device__ __forceinline__ Cell dpp_cell_from_above(const Cell &v, const Cell &edge) {
    Cell o;
    o.m = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.m), fbits(v.m), 0x130, 0xf, 0xf, false));
    o.sx = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.sx), fbits(v.sx), 0x130, 0xf, 0xf, false));
    o.sy = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.sy), fbits(v.sy), 0x130, 0xf, 0xf, false));
    o.lx = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.lx), fbits(v.lx), 0x130, 0xf, 0xf, false));
    o.ly = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.ly), fbits(v.ly), 0x130, 0xf, 0xf, false));
    o.e = __builtin_amdgcn_update_dpp(edge.e, v.e, 0x130, 0xf, 0xf, false);
    return o;
}

// One forward anti-diagonal of a stripe.  `io`: d-2 on entry, d on exit; `p1`: d-1; `carry`: the slot-below copy of
// d-2's top register (made by the step before) on entry, that of d-1 on exit; `edge`: the left stripe's last column on
// d-1 (every lane holds it, lane 0 uses it).  bx / by: X[x-1]*4 and Y[y-1]*4 of every slot.
// FLAT: every loaded model emits every base from every gap state with probability exactly 2^-2 (the shipped ones do; npr_rs.h rs_cell_emissions has
// the same switch): the four gap emissions are the constant, one LDS look-up per cell and direction instead of five.  The same factor in the same
// products: no bit changes.
template <int R, bool FLAT>
__device__ __forceinline__ void tile_emissions(const StepEnv &E, const Bases<R> &bx, const Bases<R> &by, int r, float &em, float &exs, float &exl, float &eys, float &eyl) {
    if constexpr (FLAT) {
        em = *reinterpret_cast<const float *>(E.ltab + offsetof(DevModel, em) + 5 * bx.b[r] + by.b[r]);
        exs = exl = eys = eyl = 0.25f;
    } else {
        emissions<R>(E, bx, by, r, em, exs, exl, eys, eyl);
    }
}
template <int R, bool FLAT = false>
__device__ __forceinline__ void tile_fwd_step(int d, const StepEnv &E, Diag<R> &io, const Diag<R> &p1, Cell &carry, const Cell &edge,
                                              const Bases<R> &bx, const Bases<R> &by, const Masks<R> &mk) {
    const Cell Le = dpp_cell_from_below(p1.c[R - 1], edge);  // (x-1, y) of every lane's register 0
    Diag<R> o;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float em, exs, exl, eys, eyl;
        tile_emissions<R, FLAT>(E, bx, by, r, em, exs, exl, eys, eyl);
        o.c[r] = fwd_cell<false>(E.tr, r ? p1.c[r - 1] : Le, r ? io.c[r - 1] : carry, p1.c[r], em, exs, exl, eys, eyl);
    }
    settle_diag<R>(d, o, mk);
    io = o;
    carry = Le;
}
// One backward anti-diagonal.  `io`: d+2 -> d; `s1`: d+1; `carry`: the slot-above copy of d+2's register 0 -> that of
// d+1; `edge`: the right stripe's first column on d+1 (lane 63 uses it).  bx / by: X[x]*4 and Y[y]*4 of every slot.
template <int R, bool FLAT = false>
__device__ __forceinline__ void tile_bwd_step(int d, const StepEnv &E, Diag<R> &io, const Diag<R> &s1, Cell &carry, const Cell &edge,
                                              const Bases<R> &bx, const Bases<R> &by, const Masks<R> &mk) {
    const Cell Xe = dpp_cell_from_above(s1.c[0], edge);  // (x+1, y) of every lane's top register
    Diag<R> o;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float em, exs, exl, eys, eyl;
        tile_emissions<R, FLAT>(E, bx, by, r, em, exs, exl, eys, eyl);
        o.c[r] = bwd_cell<false>(E.tr, r + 1 < R ? io.c[r + 1] : carry, r + 1 < R ? s1.c[r + 1] : Xe, s1.c[r], em, exs, exl, eys, eyl);
    }
    settle_diag<R>(d, o, mk);
    io = o;
    carry = Xe;
}

__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// PROF (NPR_TILE_PROF=1, bring-up): cycles every wavefront spends waiting -- for a neighbour's cells, for its own stores
// before it publishes, at the barriers between the sweeps -- summed into a.prof[0..3] next to its total.
template <int R, bool PROF, bool FLAT>
__global__ void __launch_bounds__(WAVE *TILE_MAX_NW) __attribute__((amdgpu_waves_per_eu(R == 2 ? 6 : 1))) k_dp_tile(KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lmodel = reinterpret_cast<float *>(smem);
    int *lmisc = reinterpret_cast<int *>(lmodel + MODEL_FLOATS);  // [0..3] totals, [4] pair counter, [5] next task
    int *prog = lmisc + 8;                                        // [TILE_MAX_NW] rows whose neighbour cells are out
    constexpr int K = 64 * R;

    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = uni(static_cast<int>(threadIdx.x) >> 6);
    const int NW = static_cast<int>(blockDim.x) >> 6;
    float *const stage = reinterpret_cast<float *>(prog + TILE_MAX_NW) + wv * (TILE_BLOCK * EDGE_FLOATS);
    char *const F = a.F + uni64(a.region[blockIdx.x]) * 8;
    const int voff = 8 * R * lane;
    int jr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) jr[r] = R * lane + r;

    uint64_t pf_spin = 0, pf_vm = 0, pf_bar = 0, pf_t0 = 0, pf_all = 0;
    auto tick = [&]() -> uint64_t { return PROF ? __builtin_readcyclecounter() : 0; };
    if constexpr (PROF) pf_t0 = tick();

    int t = blockIdx.x;
    while (t < a.ntasks) {
        const Task *tp = a.tasks + t;
        const int64_t x_off = uni64(tp->x_off), y_off = uni64(tp->y_off), pair_off = uni64(tp->pair_off),
                      tile_off = uni64(tp->tile_off), rowmask_off = uni64(tp->rowmask_off);
        const int lX = uni(tp->lX), lY = uni(tp->lY), D = uni(tp->D), pair_cap = uni(tp->pair_cap),
                  flags = uni(tp->flags), model = uni(tp->model), xs = uni(tp->xs), ys = uni(tp->ys);
        cptr32 rowmask = (cptr32)(a.rowmask + rowmask_off);  // one packed word per row, through the scalar cache
        const Stripe *tab = a.stripes + tile_off;
        const UStripe hd = load_stripe(tab, 0);
        const int S = hd.X;
        const uint32_t rows = static_cast<uint32_t>(hd.K);
        tab += 1;
        char *const Ef = F + static_cast<int64_t>(rows) * (K * 8);           // neighbour cells of the forward sweep
        char *const Eb = Ef + static_cast<int64_t>(rows) * (4 * EDGE_FLOATS);  // ... of the backward sweep
        const int rs = flags & 1, re = (flags >> 1) & 1;

        __syncthreads();
        {
            const float *gm = reinterpret_cast<const float *>(a.models + model);
            for (int i = threadIdx.x; i < MODEL_FLOATS; i += blockDim.x) lmodel[i] = gm[i];
            if (threadIdx.x == 0) lmisc[0] = 0, lmisc[1] = E_DEAD, lmisc[2] = 0, lmisc[3] = E_DEAD, lmisc[4] = 0;
            if (threadIdx.x < TILE_MAX_NW) prog[threadIdx.x] = 0;
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

        // =============================== forward ===============================
        for (int s = wv; s < S; s += NW) {
            const UStripe st = load_stripe(tab, s);
            if (st.dl >= st.df) {
            // the stripe to the left: its rows, and which wavefront sweeps it
            int dfL = 1, dlL = 0, wL = 0;
            uint32_t row0L = 0;
            if (s > 0) {
                const UStripe sl = load_stripe(tab, s - 1);
                dfL = sl.df, dlL = sl.dl, row0L = sl.row0;
                wL = (s - 1) % NW;
            }
            const int lenL = dlL - dfL + 1;
            Bases<R> bx, by;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                bx.b[r] = base4(E.X, lX, st.X + jr[r] - 1);
                by.b[r] = base4(E.Y, lY, (st.df - 1) - st.X - jr[r] - 1);  // as of anti-diagonal df - 1
            }
            Feed fy;
            feed_init<+1>(fy, E.Y, lY, st.df - st.X - 1, lane);
            Diag<R> A = dead_diag<R>(), B = dead_diag<R>();
            Cell carry = dead_cell();
            const uint64_t out_lane = 1ull << (st.K / R - 1);  // holds the stripe's last column in its top register
            int blk_lo = 0, blk_hi = 0;                         // staged cells of the left stripe: [blk_lo, blk_hi) past dfL
            cptr32 rm = rowmask + st.row0;                      // walked by pointer: the word of the row in hand
            uint32_t w_n = rm[0];                               // mask word one row ahead
            const __amdgpu_buffer_rsrc_t rsF = stripe_rsrc<R>(F, st.row0, K * 8), rsE = stripe_rsrc<R>(Ef, st.row0, 4 * EDGE_FLOATS);
            {   // (x-1, y-1) of slot 0 on the first anti-diagonal: the left stripe's cell on df - 2
                const int q0 = st.df - 2 - dfL;
                if (q0 >= 0 && q0 < lenL) {
                    const int hi = min(q0 + TILE_BLOCK, lenL);
                    const int need = static_cast<int>(row0L) + hi;
                    { const uint64_t c0 = tick(); while (uni(lds_peek(prog + wL)) < need) __builtin_amdgcn_s_sleep(2); pf_spin += tick() - c0; }
                    asm volatile("" ::: "memory");
                    edge_stage(Ef, row0L + q0, hi - q0, stage, lane);
                    blk_lo = q0, blk_hi = hi;
                    const Cell c0 = edge_get(stage, 0);
                    if (lane == 0) carry = c0;
                }
            }

            auto step = [&](int d, Diag<R> &io, const Diag<R> &p1) {
                const int k = d - st.df;
                const Masks<R> mk = row_masks(w_n);
                if (d < st.dl) w_n = rm[1];
                rm += 1;
                Cell edge = dead_cell();
                const int q = d - 1 - dfL;
                if (static_cast<unsigned>(q) < static_cast<unsigned>(lenL)) {  // uniform; 0 <= q < lenL (lenL >= 0)
                    if (q >= blk_hi) {
                        const int hi = min(q + TILE_BLOCK, lenL);
                        const int need = static_cast<int>(row0L) + hi;
                        { const uint64_t c0 = tick(); while (uni(lds_peek(prog + wL)) < need) __builtin_amdgcn_s_sleep(2); pf_spin += tick() - c0; }
                        asm volatile("" ::: "memory");
                        edge_stage(Ef, row0L + q, hi - q, stage, lane);
                        blk_lo = q, blk_hi = hi;
                    }
                    edge = edge_get(stage, q - blk_lo);
                }
                bases_down<R>(by, feed_get<+1>(fy, E.Y, lY, d - st.X - 1, lane));
                tile_fwd_step<R, FLAT>(d, E, io, p1, carry, edge, bx, by, mk);
                if (d == 0) {  // the start cell (0, 0): slot 0 of the first stripe
                    if (lane == 0) {
                        Cell c;
                        c.m = mdl->start[rs * 5 + 0], c.sx = mdl->start[rs * 5 + 1], c.sy = mdl->start[rs * 5 + 2];
                        c.lx = mdl->start[rs * 5 + 3], c.ly = mdl->start[rs * 5 + 4];
                        normalise(c, 0);
                        io.c[0] = c;
                    }
                }
                tile_store_row<R>(rsF, voff + k * (K * 8), io, mk);
                edge_store(rsE, k, io.c[R - 1], out_lane);
                if ((k & (TILE_BLOCK - 1)) == TILE_BLOCK - 1 || d == st.dl) {
                    { const uint64_t c0 = tick(); wait_vm(); pf_vm += tick() - c0; }
                    if (lane == 0) lds_poke(prog + wv, static_cast<int>(st.row0) + k + 1);
                }
            };
            int d = st.df;
            for (; d + 1 <= st.dl; d += 2) {
                step(d, B, A);
                step(d + 1, A, B);
            }
            if (d <= st.dl) step(d, B, A);
            if (s == S - 1) {  // total probability at the end corner (lX, lY), anti-diagonal D = this stripe's last row
                const bool inB = ((st.dl - st.df) & 1) == 0;
                const int je = lX - st.X;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (jr[r] == je) {
                        const Cell c = inB ? B.c[r] : A.c[r];
                        const float raw = dot5(mdl->end + re * 5, c);
                        if (raw > 0.f) {
                            int k;
                            reinterpret_cast<float *>(lmisc)[0] = __builtin_frexpf(raw, &k);
                            lmisc[1] = c.e + k;
                        }
                    }
            }
            }
        }
        { const uint64_t c0 = tick(); __syncthreads(); pf_bar += tick() - c0; }
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
            if (threadIdx.x < TILE_MAX_NW) prog[threadIdx.x] = 0x7fffffff;  // now: the LOWEST row whose neighbour cell is out
            __syncthreads();
            int s_top = S - 1 - ((S - 1 - wv) % NW + NW) % NW;  // the last stripe of this wavefront (s == wv mod NW)
            for (int s = s_top; s >= 0; s -= NW) {
                const UStripe st = load_stripe(tab, s);
                if (st.dl >= st.df) {
                int dfR = 1, dlR = 0, wR = 0;
                uint32_t row0R = 0;
                if (s + 1 < S) {
                    const UStripe sr = load_stripe(tab, s + 1);
                    dfR = sr.df, dlR = sr.dl, row0R = sr.row0;
                    wR = (s + 1) % NW;
                }
                const int lenR = dlR - dfR + 1;
                const int X0 = st.X;  // every stripe is 64*R columns wide (npr_sched.h stripe_fill): slot j is column X0 + j
                Bases<R> bx, by;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    bx.b[r] = base4(E.X, lX, X0 + jr[r]);
                    by.b[r] = base4(E.Y, lY, (st.dl + 1) - X0 - jr[r]);  // as of anti-diagonal dl + 1
                }
                Feed fy;
                feed_init<-1>(fy, E.Y, lY, st.dl - X0 - (K - 1), lane);
                Diag<R> A = dead_diag<R>(), B = dead_diag<R>();
                Cell carry = dead_cell();
                const uint64_t out_lane = 1ull;  // lane 0 holds the stripe's first column in its register 0
                int blk_lo = 0, blk_hi = 0;                     // staged cells of the right stripe: entries (blk_lo, blk_hi] ... see below
                FRow<R> fa, fb;
#pragma unroll
                for (int r = 0; r < R; ++r) fa.v[r] = fb.v[r] = 0.f, fa.e[r] = fb.e[r] = E_DEAD;
                cptr32 rm = rowmask + st.row0 + static_cast<uint32_t>(st.dl - st.df);  // walked by pointer: the word of the row in hand
                const __amdgpu_buffer_rsrc_t rsF = stripe_rsrc<R>(F, st.row0, K * 8), rsE = stripe_rsrc<R>(Eb, st.row0, 4 * EDGE_FLOATS);
                // forward row of the first anti-diagonal (the later ones are loaded one step ahead, with their masks)
                Masks<R> mk_n = row_masks(rm[0]);
                tile_load_row<R>(rsF, voff + (st.dl - st.df) * (K * 8), fb, mk_n);
                blk_lo = lenR, blk_hi = lenR;  // staged: entries [blk_lo, blk_hi) of the right stripe (entry = d' - dfR); empty
                {   // (x+1, y+1) of the top slot on the first anti-diagonal: the right stripe's cell on dl + 2
                    const int q0 = st.dl + 2 - dfR;
                    if (q0 >= 0 && q0 < lenR) {
                        const int lo = max(q0 - TILE_BLOCK + 1, 0);
                        const int need = static_cast<int>(row0R) + lo;
                        { const uint64_t c0 = tick(); while (uni(lds_peek(prog + wR)) > need) __builtin_amdgcn_s_sleep(2); pf_spin += tick() - c0; }
                        asm volatile("" ::: "memory");
                        edge_stage(Eb, row0R + lo, q0 - lo + 1, stage, lane);
                        blk_lo = lo, blk_hi = q0 + 1;
                        const Cell c0 = edge_get(stage, q0 - lo);
                        if (lane == WAVE - 1) carry = c0;
                    }
                }

                // f: the forward row of d (loaded a step ago); fnext: where the row of d-1 goes
                auto step = [&](int d, Diag<R> &io, const Diag<R> &s1, FRow<R> &f, FRow<R> &fnext) {
                    const int k = d - st.df;
                    const Masks<R> mk = mk_n;
                    if (d > st.df) {
                        rm -= 1;
                        mk_n = row_masks(rm[0]);
                        tile_load_row<R>(rsF, voff + (k - 1) * (K * 8), fnext, mk_n);
                    }
                    Cell edge = dead_cell();
                    const int q = d + 1 - dfR;
                    if (static_cast<unsigned>(q) < static_cast<unsigned>(lenR)) {  // uniform; 0 <= q < lenR (lenR >= 0)
                        if (q < blk_lo) {
                            const int lo = max(q - TILE_BLOCK + 1, 0);
                            const int need = static_cast<int>(row0R) + lo;
                            { const uint64_t c0 = tick(); while (uni(lds_peek(prog + wR)) > need) __builtin_amdgcn_s_sleep(2); pf_spin += tick() - c0; }
                            asm volatile("" ::: "memory");
                            edge_stage(Eb, row0R + lo, q - lo + 1, stage, lane);
                            blk_lo = lo, blk_hi = q + 1;
                        }
                        edge = edge_get(stage, q - blk_lo);
                    }
                    bases_up<R>(by, feed_get<-1>(fy, E.Y, lY, d - X0 - (K - 1), lane));
                    tile_bwd_step<R, FLAT>(d, E, io, s1, carry, edge, bx, by, mk);
                    if (d == D) {  // the end corner (lX, lY)
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (X0 + jr[r] == lX) {
                                Cell c;
                                c.m = mdl->end[re * 5 + 0], c.sx = mdl->end[re * 5 + 1], c.sy = mdl->end[re * 5 + 2];
                                c.lx = mdl->end[re * 5 + 3], c.ly = mdl->end[re * 5 + 4];
                                normalise(c, 0);
                                io.c[r] = c;
                            }
                    }
                    edge_store(rsE, k, io.c[0], out_lane);
                    // posteriors of this anti-diagonal, slots claimed from the workgroup's LDS counter
                    {
                        float p[R];
                        uint64_t hit[R];
                        int total = 0;
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            p[r] = posterior(f.v[r], f.e[r], io.c[r].m, io.c[r].e, tot_e, inv_tot);
                            hit[r] = __ballot(p[r] >= sink.threshold) & mk.cell[r];
                            total += __popcll(hit[r]);
                        }
                        if (d >= 2 && total) {
                            int base = 0;
                            if (lane == 0) base = atomicAdd(lmisc + 4, total);
                            base = uni(base);
                            const int y0 = d - X0;
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                if (hit[r]) {
                                    const int before = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(hit[r] >> 32),
                                                                                 __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(hit[r]), 0));
                                    const int slot = base + before;
                                    if (__builtin_amdgcn_inverse_ballot_w64(hit[r]) && slot < sink.cap) {
                                        sink.px[sink.off + slot] = X0 + jr[r] - 1 + sink.xs;
                                        sink.py[sink.off + slot] = y0 - jr[r] - 1 + sink.ys;
                                        sink.pp[sink.off + slot] = p[r];
                                    }
                                    base += __popcll(hit[r]);
                                }
                            }
                        }
                    }
                    if (((st.dl - d) & (TILE_BLOCK - 1)) == TILE_BLOCK - 1 || d == st.df) {
                        { const uint64_t c0 = tick(); wait_vm(); pf_vm += tick() - c0; }
                        if (lane == 0) lds_poke(prog + wv, static_cast<int>(st.row0) + k);
                    }
                };
                int d = st.dl;
                for (; d - 1 >= st.df; d -= 2) {
                    step(d, B, A, fb, fa);
                    step(d - 1, A, B, fa, fb);
                }
                if (d >= st.df) step(d, B, A, fb, fa);
                if (s == 0) {  // total from the backward side: the lattice point (0, 0) is the stripe's first slot on d = 0
                    const bool inB = ((st.dl - st.df) & 1) == 0;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (X0 + jr[r] == 0) {
                            const Cell cz = inB ? B.c[r] : A.c[r];
                            const float raw = dot5(mdl->start + rs * 5, cz);
                            if (raw > 0.f) {
                                int k;
                                reinterpret_cast<float *>(lmisc)[2] = __builtin_frexpf(raw, &k);
                                lmisc[3] = cz.e + k;
                            }
                        }
                }
                }
            }
            { const uint64_t c0 = tick(); __syncthreads(); pf_bar += tick() - c0; }
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
    if constexpr (PROF) {
        pf_all = tick() - pf_t0;
        if (lane == 0) {
            atomicAdd(a.prof + 0, static_cast<unsigned long long>(pf_spin));
            atomicAdd(a.prof + 1, static_cast<unsigned long long>(pf_vm));
            atomicAdd(a.prof + 2, static_cast<unsigned long long>(pf_bar));
            atomicAdd(a.prof + 3, static_cast<unsigned long long>(pf_all));
        }
    }
}


// =====================================================================================================================
// k_em_tile<2>: the Baum-Welch E-step on column stripes -- the trainer's own band (anchors, splitMatrixBiggerThanThis 300,
// nanopore/analyses/utils.py:511) is 256-310 cells across at its widest and 87 on average: too wide for one wavefront's
// frame, and on the workgroup-wide frame of k_dp_wide<2, NW, EM> one wavefront in four works.  Same sweeps as k_dp_tile; the
// forward sweep also keeps the other four states of every cell (a second row of 16 bytes per cell), and the backward
// sweep, after the cells of anti-diagonal d, adds the posterior of every transition into them to 15 per-lane accumulators
// and of every emitted symbol to per-lane bins in LDS, as k_em_stair does.  In stripe coordinates the three predecessors
// of (x, y) are the same slot on d-1, the slot below on d-1 and the slot below on d-2: the rows of d-1 and d-2 stay in
// registers (one row is loaded per step, none twice), the slot below is one DPP shift, and lane 0's comes from the left
// stripe's last column, which the forward sweep left in the task's scratch for its neighbour.
constexpr int EM_TILE_NW = 4;

template <int R>
__device__ __forceinline__ void tile_store_planes(__amdgpu_buffer_rsrc_t rs, int vo, const Diag<R> &C, const Masks<R> &mk) {
    static_assert(R == 2, "k_em_tile: two slots per lane");
    if (__builtin_amdgcn_inverse_ballot_w64(mk.lanes)) {
        __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(C.c[0].sx), fbits(C.c[0].sy), fbits(C.c[0].lx), fbits(C.c[0].ly)}, rs, vo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(C.c[1].sx), fbits(C.c[1].sy), fbits(C.c[1].lx), fbits(C.c[1].ly)}, rs, vo + 16, 0, 0);
    }
}
// all five states of a row back into registers; lanes outside the row keep what they held (every use is masked)
template <int R>
__device__ __forceinline__ void tile_load_full(__amdgpu_buffer_rsrc_t rsF, __amdgpu_buffer_rsrc_t rsX, int voF, int voX, Diag<R> &G,
                                               const Masks<R> &mk) {
    if (__builtin_amdgcn_inverse_ballot_w64(mk.lanes)) {
        const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rsF, voF, 0, 0);
        const v4i u = __builtin_amdgcn_raw_buffer_load_b128(rsX, voX, 0, 0);
        const v4i w = __builtin_amdgcn_raw_buffer_load_b128(rsX, voX + 16, 0, 0);
        G.c[0] = Cell{bitsf(q.x), bitsf(u.x), bitsf(u.y), bitsf(u.z), bitsf(u.w), q.y};
        G.c[1] = Cell{bitsf(q.z), bitsf(w.x), bitsf(w.y), bitsf(w.z), bitsf(w.w), q.w};
    }
}
// one neighbour cell (32 bytes, the same address for every lane)
__device__ __forceinline__ Cell edge_load(char *Eb, uint32_t row) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Eb + static_cast<int64_t>(row) * (4 * EDGE_FLOATS), 0, -1, 0x00020000);
    const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rs, 0, 0, 0);
    const v2i g = __builtin_amdgcn_raw_buffer_load_b64(rs, 16, 0, 0);
    return Cell{bitsf(q.x), bitsf(q.y), bitsf(q.z), bitsf(q.w), bitsf(g.x), g.y};
}

#ifndef NPR_EM_SKIP
#define NPR_EM_SKIP (-40)
#endif
constexpr int EM_SKIP = NPR_EM_SKIP;
__device__ __forceinline__ int Fm_e_of(const Cell &c) { return c.e; }
__device__ __forceinline__ float &tile_bin_at(float *lbins, int byte_off) {
    return *reinterpret_cast<float *>(reinterpret_cast<char *>(lbins) + byte_off);
}

// Expected counts of the transitions into the cells `io` of one row: Gm = (x-1, y-1), Gl = (x-1, y), Gu = (x, y-1), each with
// the lanes on which that predecessor is a band cell; eX / eY: 4 * the bases consumed into the cell (X[x-1], Y[y-1]).
// (The accumulation of k_em_stair's em_cells, npr_kernel_stair.hip, on this kernel's neighbours.)
template <int R>
__device__ __forceinline__ void tile_em_cells(const StepEnv &E, const Diag<R> &io, const Masks<R> &mk, const Diag<R> &Gm, const uint64_t (&vm)[R],
                                              const Diag<R> &Gl, const uint64_t (&vl)[R], const Diag<R> &Gu, const uint64_t (&vu)[R],
                                              const Bases<R> &eX, const Bases<R> &eY, int tot_e, float inv_tot, float (&acc)[15], float *lbins,
                                              int lane) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const Cell c = io.c[r];
        const uint64_t here = __ballot(c.e != E_DEAD) & mk.cell[r];
        const int ex4 = eX.b[r], ey4 = eY.b[r];
        const int lane4 = 4 * lane;
        float bM = 0.f, bXs = 0.f, bXl = 0.f, bYs = 0.f, bYl = 0.f;  // this cell's emission posteriors
        // (round 6) A region whose every lane's F * B / total is below 2^EM_SKIP contributes nothing a count can see (2.7e8 terms of 2^-40 each are
        // 2.4e-4 of a count): the wavefront steps over it -- a stripe row that lies off the alignment's path, a third of the rows of the trainer's band.
        const int s_m = Fm_e_of(Gm.c[r]) + c.e - tot_e, s_l = Fm_e_of(Gl.c[r]) + c.e - tot_e, s_u = Fm_e_of(Gu.c[r]) + c.e - tot_e;
        const uint64_t live_m = here & vm[r] & __ballot(s_m > EM_SKIP), live_l = here & vl[r] & __ballot(s_l > EM_SKIP), live_u = here & vu[r] & __ballot(s_u > EM_SKIP);
        if (lanes_of(live_m)) {
            const Cell &Fm = Gm.c[r];
            const int s = min(max(Fm.e + c.e - tot_e, -200), 200);
            const float em = *reinterpret_cast<const float *>(E.ltab + offsetof(DevModel, em) + 5 * ex4 + ey4);
            const float w = __builtin_ldexpf(em * c.m * inv_tot, s);
            const float t0 = Fm.m * E.tr.mm * w, t1 = Fm.sx * E.tr.sxm * w, t2 = Fm.sy * E.tr.sym * w, t3 = Fm.lx * E.tr.lxm * w,
                        t4 = Fm.ly * E.tr.lym * w;
            acc[0] += t0, acc[1] += t1, acc[2] += t2, acc[3] += t3, acc[4] += t4;
            bM = (t0 + t1) + (t2 + t3) + t4;
        }
        if (lanes_of(live_l)) {
            const Cell &Fl = Gl.c[r];
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
        if (lanes_of(live_u)) {
            const Cell &Fu = Gu.c[r];
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
        if (lanes_of(live_m | live_l | live_u)) {  // the five bins of this cell (disjoint tables): all reads, then all writes; an N base goes to the scratch row
            constexpr int TRASH = EM_BINS * 256;
            const bool nx = ex4 >= 16, ny = ey4 >= 16;
            const int aM = ((nx || ny) ? TRASH : ex4 * 256 + ey4 * 64) + lane4;
            const int aXs = (nx ? TRASH : 16 * 256 + ex4 * 64) + lane4, aXl = (nx ? TRASH : 20 * 256 + ex4 * 64) + lane4;
            const int aYs = (ny ? TRASH : 24 * 256 + ey4 * 64) + lane4, aYl = (ny ? TRASH : 28 * 256 + ey4 * 64) + lane4;
            const float v0 = tile_bin_at(lbins, aM), v1 = tile_bin_at(lbins, aXs), v2 = tile_bin_at(lbins, aXl), v3 = tile_bin_at(lbins, aYs),
                        v4 = tile_bin_at(lbins, aYl);
            tile_bin_at(lbins, aM) = v0 + bM;
            tile_bin_at(lbins, aXs) = v1 + bXs;
            tile_bin_at(lbins, aXl) = v2 + bXl;
            tile_bin_at(lbins, aYs) = v3 + bYs;
            tile_bin_at(lbins, aYl) = v4 + bYl;
        }
    }
}

#ifndef NPR_EM_TILE_WPE
#define NPR_EM_TILE_WPE 3  // wavefronts per SIMD the register allocation aims at (164 VGPRs unconstrained: 3)
#endif
template <int R>
__global__ void __launch_bounds__(WAVE *EM_TILE_NW) __attribute__((amdgpu_waves_per_eu(NPR_EM_TILE_WPE))) k_em_tile(KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lmodel = reinterpret_cast<float *>(smem);
    int *lmisc = reinterpret_cast<int *>(lmodel + MODEL_FLOATS);  // [0..3] totals, [5] next task
    int *prog = lmisc + 8;
    constexpr int K = 64 * R;

    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = uni(static_cast<int>(threadIdx.x) >> 6);
    const int NW = static_cast<int>(blockDim.x) >> 6;
    float *const stage = reinterpret_cast<float *>(prog + TILE_MAX_NW) + wv * (TILE_BLOCK * EDGE_FLOATS);
    float *const lbins = reinterpret_cast<float *>(prog + TILE_MAX_NW) + NW * (TILE_BLOCK * EDGE_FLOATS) + wv * ((EM_BINS + 1) * WAVE);
    char *const F = a.F + uni64(a.region[blockIdx.x]) * 8;
    char *const Fx = reinterpret_cast<char *>(a.Fx) + uni64(a.region[blockIdx.x]) * 16;  // the other four states: 16 bytes per cell
    const int voff = 8 * R * lane;
    int jr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) jr[r] = R * lane + r;

    int t = blockIdx.x;
    while (t < a.ntasks) {
        const Task *tp = a.tasks + t;
        const int64_t x_off = uni64(tp->x_off), y_off = uni64(tp->y_off), tile_off = uni64(tp->tile_off), rowmask_off = uni64(tp->rowmask_off);
        const int lX = uni(tp->lX), lY = uni(tp->lY), D = uni(tp->D), flags = uni(tp->flags), model = uni(tp->model);
        cptr32 rowmask = (cptr32)(a.rowmask + rowmask_off);
        const Stripe *tab = a.stripes + tile_off;
        const UStripe hd = load_stripe(tab, 0);
        const int S = hd.X;
        const uint32_t rows = static_cast<uint32_t>(hd.K);
        tab += 1;
        char *const Ef = F + static_cast<int64_t>(rows) * (K * 8);
        char *const Eb = Ef + static_cast<int64_t>(rows) * (4 * EDGE_FLOATS);
        const int rs = flags & 1, re = (flags >> 1) & 1;

        __syncthreads();
        {
            const float *gm = reinterpret_cast<const float *>(a.models + model);
            for (int i = threadIdx.x; i < MODEL_FLOATS; i += blockDim.x) lmodel[i] = gm[i];
            if (threadIdx.x == 0) lmisc[0] = 0, lmisc[1] = E_DEAD, lmisc[2] = 0, lmisc[3] = E_DEAD, lmisc[4] = 0;
            if (threadIdx.x < TILE_MAX_NW) prog[threadIdx.x] = 0;
            for (int i = 0; i <= EM_BINS; ++i) lbins[i * WAVE + lane] = 0.f;
        }
        __syncthreads();
        StepEnv E;
        E.mdl = reinterpret_cast<const DevModel *>(lmodel);
        E.ltab = reinterpret_cast<const char *>(lmodel);
        E.X = a.seq + x_off, E.Y = a.seq + y_off, E.lX = lX, E.lY = lY, E.lane = lane;
        E.tr = load_trans(E.mdl->T);  // in VGPRs, as in k_em_stair: every count multiplies by one
        const DevModel *mdl = E.mdl;

        // =============================== forward: as k_dp_tile, all five states stored ===============================
        for (int s = wv; s < S; s += NW) {
            const UStripe st = load_stripe(tab, s);
            if (st.dl >= st.df) {
            int dfL = 1, dlL = 0, wL = 0;
            uint32_t row0L = 0;
            if (s > 0) {
                const UStripe sl = load_stripe(tab, s - 1);
                dfL = sl.df, dlL = sl.dl, row0L = sl.row0;
                wL = (s - 1) % NW;
            }
            const int lenL = dlL - dfL + 1;
            Bases<R> bx, by;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                bx.b[r] = base4(E.X, lX, st.X + jr[r] - 1);
                by.b[r] = base4(E.Y, lY, (st.df - 1) - st.X - jr[r] - 1);
            }
            Feed fy;
            feed_init<+1>(fy, E.Y, lY, st.df - st.X - 1, lane);
            Diag<R> A = dead_diag<R>(), B = dead_diag<R>();
            Cell carry = dead_cell();
            const uint64_t out_lane = 1ull << (st.K / R - 1);
            int blk_lo = 0, blk_hi = 0;
            cptr32 rm = rowmask + st.row0;
            uint32_t w_n = rm[0];
            const __amdgpu_buffer_rsrc_t rsF = stripe_rsrc<R>(F, st.row0, K * 8), rsE = stripe_rsrc<R>(Ef, st.row0, 4 * EDGE_FLOATS),
                                         rsX = stripe_rsrc<R>(Fx, st.row0, K * 16);
            {
                const int q0 = st.df - 2 - dfL;
                if (q0 >= 0 && q0 < lenL) {
                    const int hi = min(q0 + TILE_BLOCK, lenL);
                    const int need = static_cast<int>(row0L) + hi;
                    while (uni(lds_peek(prog + wL)) < need) __builtin_amdgcn_s_sleep(2);
                    asm volatile("" ::: "memory");
                    edge_stage(Ef, row0L + q0, hi - q0, stage, lane);
                    blk_lo = q0, blk_hi = hi;
                    const Cell c0 = edge_get(stage, 0);
                    if (lane == 0) carry = c0;
                }
            }
            auto step = [&](int d, Diag<R> &io, const Diag<R> &p1) {
                const int k = d - st.df;
                const Masks<R> mk = row_masks(w_n);
                if (d < st.dl) w_n = rm[1];
                rm += 1;
                Cell edge = dead_cell();
                const int q = d - 1 - dfL;
                if (static_cast<unsigned>(q) < static_cast<unsigned>(lenL)) {
                    if (q >= blk_hi) {
                        const int hi = min(q + TILE_BLOCK, lenL);
                        const int need = static_cast<int>(row0L) + hi;
                        while (uni(lds_peek(prog + wL)) < need) __builtin_amdgcn_s_sleep(2);
                        asm volatile("" ::: "memory");
                        edge_stage(Ef, row0L + q, hi - q, stage, lane);
                        blk_lo = q, blk_hi = hi;
                    }
                    edge = edge_get(stage, q - blk_lo);
                }
                bases_down<R>(by, feed_get<+1>(fy, E.Y, lY, d - st.X - 1, lane));
                tile_fwd_step<R>(d, E, io, p1, carry, edge, bx, by, mk);
                if (d == 0) {
                    if (lane == 0) {
                        Cell c;
                        c.m = mdl->start[rs * 5 + 0], c.sx = mdl->start[rs * 5 + 1], c.sy = mdl->start[rs * 5 + 2];
                        c.lx = mdl->start[rs * 5 + 3], c.ly = mdl->start[rs * 5 + 4];
                        normalise(c, 0);
                        io.c[0] = c;
                    }
                }
                tile_store_row<R>(rsF, voff + k * (K * 8), io, mk);
                tile_store_planes<R>(rsX, 2 * voff + k * (K * 16), io, mk);
                edge_store(rsE, k, io.c[R - 1], out_lane);
                if ((k & (TILE_BLOCK - 1)) == TILE_BLOCK - 1 || d == st.dl) {
                    wait_vm();
                    if (lane == 0) lds_poke(prog + wv, static_cast<int>(st.row0) + k + 1);
                }
            };
            int d = st.df;
            for (; d + 1 <= st.dl; d += 2) {
                step(d, B, A);
                step(d + 1, A, B);
            }
            if (d <= st.dl) step(d, B, A);
            if (s == S - 1) {
                const bool inB = ((st.dl - st.df) & 1) == 0;
                const int je = lX - st.X;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (jr[r] == je) {
                        const Cell c = inB ? B.c[r] : A.c[r];
                        const float raw = dot5(mdl->end + re * 5, c);
                        if (raw > 0.f) {
                            int k;
                            reinterpret_cast<float *>(lmisc)[0] = __builtin_frexpf(raw, &k);
                            lmisc[1] = c.e + k;
                        }
                    }
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

        // =============================== backward + expected counts ===============================
        float acc[15];
#pragma unroll
        for (int i = 0; i < 15; ++i) acc[i] = 0.f;
        if (alive) {
            const float inv_tot = 1.0f / tot_m;
            if (threadIdx.x < TILE_MAX_NW) prog[threadIdx.x] = 0x7fffffff;
            __syncthreads();
            int s_top = S - 1 - ((S - 1 - wv) % NW + NW) % NW;
            for (int s = s_top; s >= 0; s -= NW) {
                const UStripe st = load_stripe(tab, s);
                if (st.dl >= st.df) {
                int dfR = 1, dlR = 0, wR = 0;
                uint32_t row0R = 0;
                if (s + 1 < S) {
                    const UStripe sr = load_stripe(tab, s + 1);
                    dfR = sr.df, dlR = sr.dl, row0R = sr.row0;
                    wR = (s + 1) % NW;
                }
                const int lenR = dlR - dfR + 1;
                int dfL = 1, dlL = 0;
                uint32_t row0L = 0;
                if (s > 0) {
                    const UStripe sl = load_stripe(tab, s - 1);
                    dfL = sl.df, dlL = sl.dl, row0L = sl.row0;
                }
                const int lenL = dlL - dfL + 1;
                const int X0 = st.X;
                Bases<R> bx, by, bxm, bym;  // X[x], Y[y] of every slot (the backward cell) and X[x-1], Y[y-1] (the symbols emitted INTO it)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    bx.b[r] = base4(E.X, lX, X0 + jr[r]);
                    bxm.b[r] = base4(E.X, lX, X0 + jr[r] - 1);
                    by.b[r] = base4(E.Y, lY, (st.dl + 1) - X0 - jr[r]);
                    bym.b[r] = base4(E.Y, lY, (st.dl + 1) - X0 - jr[r] - 1);
                }
                Feed fy, fym;
                feed_init<-1>(fy, E.Y, lY, st.dl - X0 - (K - 1), lane);
                feed_init<-1>(fym, E.Y, lY, st.dl - X0 - (K - 1) - 1, lane);
                Diag<R> A = dead_diag<R>(), B = dead_diag<R>();
                Cell carry = dead_cell();
                const uint64_t out_lane = 1ull;
                int blk_lo = lenR, blk_hi = lenR;
                cptr32 rm0 = rowmask + st.row0;
                const __amdgpu_buffer_rsrc_t rsF = stripe_rsrc<R>(F, st.row0, K * 8), rsE = stripe_rsrc<R>(Eb, st.row0, 4 * EDGE_FLOATS),
                                             rsX = stripe_rsrc<R>(Fx, st.row0, K * 16);
                // forward rows of d-1 and d-2 (all five states), the masks of their lanes, and the left stripe's last column on them
                Diag<R> G1 = dead_diag<R>(), G2 = dead_diag<R>();
                Masks<R> m1{}, m2{};
                Cell eL1 = dead_cell(), eL2 = dead_cell();
                auto fetch_row = [&](int dd, Diag<R> &G, Masks<R> &m, Cell &eL) {
                    m.cell[0] = m.cell[1] = m.lanes = 0, m.l0 = 0;
                    if (dd >= st.df && dd <= st.dl) {  // uniform
                        const int k = dd - st.df;
                        m = row_masks(rm0[k]);
                        tile_load_full<R>(rsF, rsX, voff + k * (K * 8), 2 * voff + k * (K * 16), G, m);
                    }
                    eL = dead_cell();
                    const int q = dd - dfL;
                    if (static_cast<unsigned>(q) < static_cast<unsigned>(lenL)) eL = edge_load(Ef, row0L + static_cast<uint32_t>(q));
                };
                fetch_row(st.dl - 1, G1, m1, eL1);
                fetch_row(st.dl - 2, G2, m2, eL2);
                Masks<R> mk_n = row_masks(rm0[st.dl - st.df]);
                {
                    const int q0 = st.dl + 2 - dfR;
                    if (q0 >= 0 && q0 < lenR) {
                        const int lo = max(q0 - TILE_BLOCK + 1, 0);
                        const int need = static_cast<int>(row0R) + lo;
                        while (uni(lds_peek(prog + wR)) > need) __builtin_amdgcn_s_sleep(2);
                        asm volatile("" ::: "memory");
                        edge_stage(Eb, row0R + lo, q0 - lo + 1, stage, lane);
                        blk_lo = lo, blk_hi = q0 + 1;
                        const Cell c0 = edge_get(stage, q0 - lo);
                        if (lane == WAVE - 1) carry = c0;
                    }
                }
                auto step = [&](int d, Diag<R> &io, const Diag<R> &s1) {
                    const int k = d - st.df;
                    const Masks<R> mk = mk_n;
                    if (d > st.df) mk_n = row_masks(rm0[k - 1]);
                    Cell edge = dead_cell();
                    const int q = d + 1 - dfR;
                    if (static_cast<unsigned>(q) < static_cast<unsigned>(lenR)) {
                        if (q < blk_lo) {
                            const int lo = max(q - TILE_BLOCK + 1, 0);
                            const int need = static_cast<int>(row0R) + lo;
                            while (uni(lds_peek(prog + wR)) > need) __builtin_amdgcn_s_sleep(2);
                            asm volatile("" ::: "memory");
                            edge_stage(Eb, row0R + lo, q - lo + 1, stage, lane);
                            blk_lo = lo, blk_hi = q + 1;
                        }
                        edge = edge_get(stage, q - blk_lo);
                    }
                    bases_up<R>(by, feed_get<-1>(fy, E.Y, lY, d - X0 - (K - 1), lane));
                    bases_up<R>(bym, feed_get<-1>(fym, E.Y, lY, d - X0 - (K - 1) - 1, lane));
                    tile_bwd_step<R>(d, E, io, s1, carry, edge, bx, by, mk);
                    if (d == D) {
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (X0 + jr[r] == lX) {
                                Cell c;
                                c.m = mdl->end[re * 5 + 0], c.sx = mdl->end[re * 5 + 1], c.sy = mdl->end[re * 5 + 2];
                                c.lx = mdl->end[re * 5 + 3], c.ly = mdl->end[re * 5 + 4];
                                normalise(c, 0);
                                io.c[r] = c;
                            }
                    }
                    edge_store(rsE, k, io.c[0], out_lane);
                    if (d >= 1) {  // expected counts of the transitions into the cells of d
                        Diag<R> Gl, Gm;
                        Gl.c[1] = G1.c[0], Gl.c[0] = dpp_cell_from_below(G1.c[1], eL1);
                        Gm.c[1] = G2.c[0], Gm.c[0] = dpp_cell_from_below(G2.c[1], eL2);
                        const uint64_t vu[R] = {m1.cell[0], m1.cell[1]};
                        const uint64_t vl[R] = {(m1.cell[1] << 1) | (uni(eL1.e) != E_DEAD ? 1ull : 0ull), m1.cell[0]};
                        const uint64_t vm[R] = {(m2.cell[1] << 1) | (uni(eL2.e) != E_DEAD ? 1ull : 0ull), m2.cell[0]};
                        tile_em_cells<R>(E, io, mk, Gm, vm, Gl, vl, G1, vu, bxm, bym, tot_e, inv_tot, acc, lbins, lane);
                    }
                    // the rows the next anti-diagonal needs: d-2 moves up, d-3 comes from memory
                    G1 = G2, m1 = m2, eL1 = eL2;
                    fetch_row(d - 3, G2, m2, eL2);
                    if (((st.dl - d) & (TILE_BLOCK - 1)) == TILE_BLOCK - 1 || d == st.df) {
                        wait_vm();
                        if (lane == 0) lds_poke(prog + wv, static_cast<int>(st.row0) + k);
                    }
                };
                int d = st.dl;
                for (; d - 1 >= st.df; d -= 2) {
                    step(d, B, A);
                    step(d - 1, A, B);
                }
                if (d >= st.df) step(d, B, A);
                if (s == 0) {
                    const bool inB = ((st.dl - st.df) & 1) == 0;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (X0 + jr[r] == 0) {
                            const Cell cz = inB ? B.c[r] : A.c[r];
                            const float raw = dot5(mdl->start + rs * 5, cz);
                            if (raw > 0.f) {
                                int k;
                                reinterpret_cast<float *>(lmisc)[2] = __builtin_frexpf(raw, &k);
                                lmisc[3] = cz.e + k;
                            }
                        }
                }
                }
            }
            __syncthreads();
            out.btot_m = unif(reinterpret_cast<float *>(lmisc)[2]);
            out.btot_e = uni(lmisc[3]);
            // every wavefront adds up its own bins, then its transition accumulators through the same rows
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
        }
        if (threadIdx.x == 0) {
            a.outs[t] = out;
            lmisc[5] = atomicAdd(a.queue, 1);
        }
        __syncthreads();
        t = uni(lmisc[5]) + static_cast<int>(gridDim.x);
    }
}

}  // namespace

size_t tile_lds_bytes(int nw) { return sizeof(float) * (MODEL_FLOATS + 8 + TILE_MAX_NW + static_cast<size_t>(nw) * TILE_BLOCK * EDGE_FLOATS); }

// rows of 64*R cells plus one 32-byte neighbour cell per row for each sweep direction
int64_t tile_scratch_cells(int64_t rows, int R) { return rows * (64 * R + 2 * EDGE_FLOATS / 2); }

size_t em_tile_lds_bytes(int nw) { return tile_lds_bytes(nw) + sizeof(float) * static_cast<size_t>(nw) * (EM_BINS + 1) * WAVE; }
int em_tile_waves_per_cu() { return 4 * NPR_EM_TILE_WPE; }
int em_tile_waves() {  // wavefronts per task
    return 2;  // trainer's band 2.24 / 2.53 / 2.40 / 2.16e10 cells/s on 1 / 2 / 3 / 4 wavefronts per task, a 560-cell band 2.6 / 3.3 / 3.1 / 3.3e10
}

int launch_em_tile(const KernelArgs &a, int R, int grid, void *stream) {
    if (R != 2) return static_cast<int>(hipErrorInvalidValue);
    const int nw = em_tile_waves();
    hipLaunchKernelGGL((k_em_tile<2>), dim3(grid), dim3(WAVE * nw), em_tile_lds_bytes(nw), static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}

int launch_tile(const KernelArgs &a, int R, int NW, int grid, void *stream, bool flat) {  // flat: every loaded model's gap emissions are exactly 2^-2
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (NW < 1 || NW > TILE_MAX_NW) return static_cast<int>(hipErrorInvalidValue);
    const size_t lds = tile_lds_bytes(NW);
    if (R == 2 && a.prof)
        hipLaunchKernelGGL((k_dp_tile<2, true, false>), dim3(grid), dim3(WAVE * NW), lds, s, a);
    else if (R == 2 && flat)
        hipLaunchKernelGGL((k_dp_tile<2, false, true>), dim3(grid), dim3(WAVE * NW), lds, s, a);
    else if (R == 2)
        hipLaunchKernelGGL((k_dp_tile<2, false, false>), dim3(grid), dim3(WAVE * NW), lds, s, a);
    else
        return static_cast<int>(hipErrorInvalidValue);
    return static_cast<int>(hipGetLastError());
}

}  // namespace npr
