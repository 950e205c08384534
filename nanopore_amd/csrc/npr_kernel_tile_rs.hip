// npr_kernel_tile_rs.hip -- k_dp_tile_rs: the column-stripe kernel for WIDE bands (k_dp_tile's mapping: npr_kernel_tile.hip) in
// ROW-SCALED arithmetic (npr_rs.h): one exponent per anti-diagonal row of a STRIPE -- of one wavefront --
// instead of one per cell.
//
// Same recurrences -- cactus_realign's banded five-state forward / backward / posterior pass, SURVEY.md 8a rows a5.3-a5.5,
// reference call site nanopore/analyses/utils.py:587, for the band the reference's own parameters give (anchors +-
// diagonalExpansion 10, 14 trimmed columns, splitMatrixBiggerThanThis 3000) --, same stripes, same stripe tables and row
// masks, same pipeline of wavefronts without a barrier, same outputs.  What differs from k_dp_tile:
//   * a cell is five plain fp32 values relative to 2^e, e one scalar per wavefront: the rows of d-1 and d-2 held in registers
//     and the carried neighbour copy share it.  After every 16th anti-diagonal the largest of them is brought to 2^85
//     (rs_renorm);
//   * the cell a stripe hands to its neighbour travels with its row's exponent (the 32-byte record had room for it all
//     along); the receiver rescales it to its own, and when what arrives would land above 2^105 in its own scale -- the
//     alignment enters the stripe -- it first moves its own rows down;
//   * slots outside the band are computed like any other, with the dead base code as their bases: all emissions 0, so every
//     state an exact zero (npr_rs.h RS_DEAD8) -- no exponent to mark a dead cell, no EXEC mask;
//   * a forward row is 4 bytes per cell in the scratch; its exponent is the one in the neighbour record of the same row, read
//     back through the scalar cache;
//   * the task keeps the range certificate of k_dp_rs (npr_device.h, with 2^20 more headroom for what a neighbour may hand
//     over): a task without it is run again by npr_batch_run with k_dp_tile.
// Scaling by powers of two is exact, so wherever nothing leaves fp32's range relative to its row the results are those of the
// per-cell-exponent kernels bit for bit, and the parity tests compare them with that mirror.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "npr_device.h"
#include "npr_frame.h"
#include "npr_rs.h"

namespace npr {

namespace {

constexpr int TRS_MAX_NW = 8;   // wavefronts per workgroup (launch bound)
constexpr int TRS_BLOCK = 16;   // neighbour cells staged / published at a time
constexpr int TRS_EDGE = 8;     // one neighbour cell in memory: m, sx, sy, lx, ly, e, -, -  (k_dp_tile's record)
constexpr int TRS_ADAPT = 20;   // a neighbour cell that would land more than this many binary orders above 2^NPR_RS_TOP moves the own rows down
constexpr int TRS_E_NONE = -(1 << 24);  // exponent of a stripe nothing has entered yet
// the certificate of k_dp_rs with TRS_ADAPT more headroom for the rows' maxima (npr_device.h NPR_RS_S_LIMIT)
constexpr int TRS_S_LIMIT = 126 - 60 - (NPR_RS_TOP + 6 + TRS_ADAPT) - 1;

typedef const __attribute__((address_space(4))) int32_t *cptr_i32;

struct UStripe {
    int X, K, df, dl;
    uint32_t row0;
};
__device__ __forceinline__ UStripe load_stripe(const Stripe *tab, int s) {
    cptr_i32 p = (cptr_i32)(tab + s);
    return UStripe{p[0], p[1], p[2], p[3], static_cast<uint32_t>(p[4])};
}
__device__ __forceinline__ Masks<2> row_masks(uint32_t w) {
    Masks<2> m;
    m.cell[0] = (~0ull << (w & 63u)) & (~0ull >> ((w >> 6) & 63u));
    m.cell[1] = (~0ull << ((w >> 12) & 63u)) & (~0ull >> ((w >> 18) & 63u));
    m.lanes = m.cell[0] | m.cell[1];
    m.l0 = 0;
    return m;
}
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ int lds_peek(const int *p) { return *(const volatile lds_int *)(p); }
__device__ __forceinline__ void lds_poke(int *p, int v) { *(volatile lds_int *)(p) = v; }

// One row per anti-diagonal of a stripe: 128 cells of 4 bytes, lane l at 8 l (the region is laid out for k_dp_tile's 8-byte
// cells: this kernel uses the first half of every row's space, rows at the same row stride).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stripe_rsrc(char *base, uint32_t row0, int row_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base + static_cast<int64_t>(row0) * row_bytes, 0, -1, 0x00020000);
}
__device__ __forceinline__ void trs_store_row(__amdgpu_buffer_rsrc_t rs, int vo, const RDiag<2> &C, const Masks<2> &mk) {
    if (lanes_of(mk.lanes)) __builtin_amdgcn_raw_buffer_store_b64(v2i{fbits(C.c[0].m), fbits(C.c[1].m)}, rs, vo, 0, 0);
}
__device__ __forceinline__ void trs_load_row(__amdgpu_buffer_rsrc_t rs, int vo, RFRow<2> &f, const Masks<2> &mk) {
    if (lanes_of(mk.lanes)) {
        const v2i q = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, 0, 0);
        f.v[0] = bitsf(q.x), f.v[1] = bitsf(q.y);
    }
}
// the neighbour cell of a row with its row's exponent: 32 bytes at (stripe's first) + 32 * (row - first row)
__device__ __forceinline__ void trs_edge_store(__amdgpu_buffer_rsrc_t rs, int k, const RCell &c, int e, uint64_t lane_mask) {
    if (lanes_of(lane_mask)) {
        const int vo = 4 * TRS_EDGE * k;
        __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(c.m), fbits(c.sx), fbits(c.sy), fbits(c.lx)}, rs, vo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(v2i{fbits(c.ly), e}, rs, vo + 16, 0, 0);
    }
}
// `cnt` neighbour cells starting at row `row` into this wavefront's LDS staging (lane l takes cell l); the loads bypass the
// vector L1 (sc1): the producer is another wavefront of this workgroup
__device__ __forceinline__ void trs_edge_stage(char *Eb, uint32_t row, int cnt, float *stage, int lane) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Eb + static_cast<int64_t>(row) * (4 * TRS_EDGE), 0, -1, 0x00020000);
    if (lanes_of(low_lanes(cnt))) {
        const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rs, 32 * lane, 0, 16);
        const v2i g = __builtin_amdgcn_raw_buffer_load_b64(rs, 32 * lane + 16, 0, 16);
        *reinterpret_cast<v4i *>(stage + TRS_EDGE * lane) = q;
        *reinterpret_cast<v2i *>(stage + TRS_EDGE * lane + 4) = g;
    }
}
struct EdgeCell {
    RCell c;
    int e;
};
__device__ __forceinline__ EdgeCell trs_edge_get(const float *stage, int k) {
    const float4 q = *reinterpret_cast<const float4 *>(stage + TRS_EDGE * k);
    const float2 g = *reinterpret_cast<const float2 *>(stage + TRS_EDGE * k + 4);
    return EdgeCell{RCell{q.x, q.y, q.z, q.w, g.x}, uni(fbits(g.y))};
}

__device__ __forceinline__ RCell dpp_rcell_from_below(const RCell &v, const RCell &edge) {
    RCell o;
    o.m = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.m), fbits(v.m), 0x138, 0xf, 0xf, false));
    o.sx = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.sx), fbits(v.sx), 0x138, 0xf, 0xf, false));
    o.sy = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.sy), fbits(v.sy), 0x138, 0xf, 0xf, false));
    o.lx = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.lx), fbits(v.lx), 0x138, 0xf, 0xf, false));
    o.ly = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.ly), fbits(v.ly), 0x138, 0xf, 0xf, false));
    return o;
}
__device__ __forceinline__ RCell dpp_rcell_from_above(const RCell &v, const RCell &edge) {
    RCell o;
    o.m = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.m), fbits(v.m), 0x130, 0xf, 0xf, false));
    o.sx = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.sx), fbits(v.sx), 0x130, 0xf, 0xf, false));
    o.sy = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.sy), fbits(v.sy), 0x130, 0xf, 0xf, false));
    o.lx = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.lx), fbits(v.lx), 0x130, 0xf, 0xf, false));
    o.ly = bitsf(__builtin_amdgcn_update_dpp(fbits(edge.ly), fbits(v.ly), 0x130, 0xf, 0xf, false));
    return o;
}

// A stripe's register state: the rows of the two previous anti-diagonals, the carried neighbour copy, their common exponent.
struct TrsState {
    RDiag<2> A, B;
    RCell carry;
    int e;
};
__device__ __forceinline__ void trs_scale(TrsState &Q, float f) {
#pragma unroll
    for (int r = 0; r < 2; ++r) rcell_scale(Q.A.c[r], f), rcell_scale(Q.B.c[r], f);
    rcell_scale(Q.carry, f);
}
// The neighbour's cell in the own scale.  Its exponent alone says little -- the neighbour's rows may be large far away from
// the column it hands over --, so the decision is made on the cell's largest value: when that would land above 2^105 in the
// own scale (the alignment enters this stripe, or the stripe is still empty) the own rows move down first, so that it lands
// at 2^84: what they hold is that much smaller than what is coming in.
__device__ __forceinline__ RCell trs_take_edge(TrsState &Q, const EdgeCell &ed) {
    const int xe = static_cast<int>(uni(static_cast<int>(rcell_max_bits(ed.c))) >> 23);  // biased exponent of the largest value (uniform)
    if (xe == 0) return zero_rcell();  // nothing (or less than fp32 can hold) comes in
    int k = ed.e - Q.e;
    const int fin = xe + k;            // its biased exponent in the own scale
    if (fin > 127 + NPR_RS_TOP + TRS_ADAPT) {  // uniform, rare
        const int sh = fin - (127 + NPR_RS_TOP - 1);
        trs_scale(Q, sh > 126 ? 0.f : bitsf((127 - sh) << 23));  // 2^-sh; further than fp32 reaches: nothing is left
        Q.e += sh;
        k -= sh;
    }
    RCell c = ed.c;
    c.m = __builtin_ldexpf(c.m, k), c.sx = __builtin_ldexpf(c.sx, k), c.sy = __builtin_ldexpf(c.sy, k);
    c.lx = __builtin_ldexpf(c.lx, k), c.ly = __builtin_ldexpf(c.ly, k);
    return c;
}
// renormalisation of everything a stripe holds (rs_renorm's rule over the two rows and the carried copy)
__device__ __forceinline__ void trs_renorm(TrsState &Q) {
    uint32_t u = rcell_max_bits(Q.carry);
#pragma unroll
    for (int r = 0; r < 2; ++r) u = umax3(u, rcell_max_bits(Q.A.c[r]), rcell_max_bits(Q.B.c[r]));
    const uint32_t top = wave_max_u32(u);
    const int eb = static_cast<int>(top >> 23);
    if (eb == 0) return;
    const int k = min(max(RS_TOP + 126 - eb, -126), 127);
    trs_scale(Q, bitsf((k + 127) << 23));
    Q.e -= k;
}

__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#ifndef NPR_TRS_WAVES
#define NPR_TRS_WAVES 6
#endif
#ifndef NPR_TRS_T_SGPR
#define NPR_TRS_T_SGPR 1
#endif
__global__ void __launch_bounds__(WAVE *TRS_MAX_NW) __attribute__((amdgpu_waves_per_eu(NPR_TRS_WAVES))) k_dp_tile_rs(KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RsTables *ltab = reinterpret_cast<RsTables *>(smem);
    float *lmodel = reinterpret_cast<float *>(smem) + ((RS_TABLE_FLOATS + 3) & ~3);
    int *lmisc = reinterpret_cast<int *>(lmodel + MODEL_FLOATS);  // [0..3] totals, [4] pair counter, [5] next task, [6] largest s
    int *prog = lmisc + 8;                                        // [TRS_MAX_NW] rows whose neighbour cells are out
    constexpr int R = 2, K = 64 * R;

    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = uni(static_cast<int>(threadIdx.x) >> 6);
    const int NW = static_cast<int>(blockDim.x) >> 6;
    float *const stage = reinterpret_cast<float *>(prog + TRS_MAX_NW) + wv * (TRS_BLOCK * TRS_EDGE);
    char *const F = a.F + uni64(a.region[blockIdx.x]) * 8;
    const int voff = 4 * R * lane;
    int jr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) jr[r] = R * lane + r;

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
        char *const Ef = F + static_cast<int64_t>(rows) * (K * 8);          // neighbour cells of the forward sweep (+ the rows' exponents)
        char *const Eb = Ef + static_cast<int64_t>(rows) * (4 * TRS_EDGE);  // ... of the backward sweep
        const int rs = flags & 1, re = (flags >> 1) & 1;

        __syncthreads();
        {
            const float *gm = reinterpret_cast<const float *>(a.models + model);
            for (int i = threadIdx.x; i < MODEL_FLOATS; i += blockDim.x) lmodel[i] = gm[i];
            if (threadIdx.x == 0) lmisc[0] = 0, lmisc[1] = E_DEAD, lmisc[2] = 0, lmisc[3] = E_DEAD, lmisc[4] = 0, lmisc[6] = -(1 << 30);
            if (threadIdx.x < TRS_MAX_NW) prog[threadIdx.x] = 0;
        }
        __syncthreads();
        rs_build_tables(ltab, reinterpret_cast<const DevModel *>(lmodel), threadIdx.x, blockDim.x);
        __syncthreads();
        StepEnv E;
        E.mdl = reinterpret_cast<const DevModel *>(lmodel);
        E.ltab = reinterpret_cast<const char *>(ltab);
        E.X = a.seq + x_off, E.Y = a.seq + y_off, E.lX = lX, E.lY = lY, E.lane = lane;
        {
            Trans tr = load_trans(E.mdl->T);
#if NPR_TRS_T_SGPR
            tr.mm = unif(tr.mm), tr.sxm = unif(tr.sxm), tr.sym = unif(tr.sym), tr.lxm = unif(tr.lxm), tr.lym = unif(tr.lym);
            tr.msx = unif(tr.msx), tr.sxsx = unif(tr.sxsx), tr.sysx = unif(tr.sysx);
            tr.msy = unif(tr.msy), tr.sysy = unif(tr.sysy), tr.sxsy = unif(tr.sxsy);
            tr.mlx = unif(tr.mlx), tr.lxlx = unif(tr.lxlx), tr.mly = unif(tr.mly), tr.lyly = unif(tr.lyly);
#endif
            E.tr = tr;
        }
        const DevModel *mdl = E.mdl;

        // =============================== forward ===============================
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
                bx.b[r] = base8<RS_XS>(E.X, lX, st.X + jr[r] - 1);
                by.b[r] = base8(E.Y, lY, (st.df - 1) - st.X - jr[r] - 1);  // as of anti-diagonal df - 1
            }
            Feed fy;
            feed8_init<+1>(fy, E.Y, lY, st.df - st.X - 1, lane);
            TrsState Q;
            Q.A = zero_rdiag<R>(), Q.B = zero_rdiag<R>(), Q.carry = zero_rcell(), Q.e = TRS_E_NONE;
            const uint64_t out_lane = 1ull << (st.K / R - 1);  // holds the stripe's last column in its top register
            int blk_lo = 0, blk_hi = 0;                         // staged cells of the left stripe: [blk_lo, blk_hi) past dfL
            cptr32 rm = rowmask + st.row0;
            uint32_t w_n = rm[0];
            const __amdgpu_buffer_rsrc_t rsF = stripe_rsrc(F, st.row0, K * 8), rsE = stripe_rsrc(Ef, st.row0, 4 * TRS_EDGE);
            {   // (x-1, y-1) of slot 0 on the first anti-diagonal: the left stripe's cell on df - 2
                const int q0 = st.df - 2 - dfL;
                if (q0 >= 0 && q0 < lenL) {
                    const int hi = min(q0 + TRS_BLOCK, lenL);
                    const int need = static_cast<int>(row0L) + hi;
                    while (uni(lds_peek(prog + wL)) < need) __builtin_amdgcn_s_sleep(2);
                    asm volatile("" ::: "memory");
                    trs_edge_stage(Ef, row0L + q0, hi - q0, stage, lane);
                    blk_lo = q0, blk_hi = hi;
                    const RCell c0 = trs_take_edge(Q, trs_edge_get(stage, 0));
                    if (lane == 0) Q.carry = c0;
                }
            }

            auto step = [&](int d, RDiag<R> &io, const RDiag<R> &p1) {
                const int k = d - st.df;
                const Masks<R> mk = row_masks(w_n);
                if (d < st.dl) w_n = rm[1];
                rm += 1;
                RCell edge = zero_rcell();
                const int q = d - 1 - dfL;
                if (static_cast<unsigned>(q) < static_cast<unsigned>(lenL)) {  // uniform
                    if (q >= blk_hi) {
                        const int hi = min(q + TRS_BLOCK, lenL);
                        const int need = static_cast<int>(row0L) + hi;
                        while (uni(lds_peek(prog + wL)) < need) __builtin_amdgcn_s_sleep(2);
                        asm volatile("" ::: "memory");
                        trs_edge_stage(Ef, row0L + q, hi - q, stage, lane);
                        blk_lo = q, blk_hi = hi;
                    }
                    edge = trs_take_edge(Q, trs_edge_get(stage, q - blk_lo));
                }
                bases_down<R>(by, feed8_get<+1>(fy, E.Y, lY, d - st.X - 1, lane));
                // one forward anti-diagonal: io d-2 -> d; p1 d-1; carry: the slot-below copy of d-2's top register -> that of d-1
                const RCell Le = dpp_rcell_from_below(p1.c[R - 1], edge);  // (x-1, y) of every lane's register 0
                RDiag<R> o;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float em, exs, exl, eys, eyl;
                    rs_emissions(E.ltab, lanes_of(mk.cell[r]) ? bx.b[r] : RS_DEADX, lanes_of(mk.cell[r]) ? by.b[r] : RS_DEAD8, em, exs, exl, eys, eyl);
                    o.c[r] = rs_fwd_cell(E.tr, r ? p1.c[r - 1] : Le, r ? io.c[r - 1] : Q.carry, p1.c[r], em, exs, exl, eys, eyl);
                }
                io = o;
                Q.carry = Le;
                if (d == 0) {  // the start cell (0, 0): slot 0 of the first stripe
                    if (lane == 0) {
                        RCell c;
                        c.m = mdl->start[rs * 5 + 0], c.sx = mdl->start[rs * 5 + 1], c.sy = mdl->start[rs * 5 + 2];
                        c.lx = mdl->start[rs * 5 + 3], c.ly = mdl->start[rs * 5 + 4];
                        io.c[0] = c;
                    }
                    Q.e = 0;
                }
                if ((d & (RS_K - 1)) == 0 && d > 0) trs_renorm(Q);
                trs_store_row(rsF, voff + k * (K * 8), io, mk);
                trs_edge_store(rsE, k, io.c[R - 1], Q.e, out_lane);
                if ((k & (TRS_BLOCK - 1)) == TRS_BLOCK - 1 || d == st.dl) {
                    wait_vm();
                    if (lane == 0) lds_poke(prog + wv, static_cast<int>(st.row0) + k + 1);
                }
            };
            int d = st.df;
            for (; d + 1 <= st.dl; d += 2) {
                step(d, Q.B, Q.A);
                step(d + 1, Q.A, Q.B);
            }
            if (d <= st.dl) step(d, Q.B, Q.A);
            if (s == S - 1) {  // total probability at the end corner (lX, lY), anti-diagonal D = this stripe's last row
                const bool inB = ((st.dl - st.df) & 1) == 0;
                const int je = lX - st.X;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (jr[r] == je) {
                        const RCell c = inB ? Q.B.c[r] : Q.A.c[r];
                        const float raw = rs_dot5(mdl->end + re * 5, c);
                        if (raw > 0.f) {
                            int k;
                            reinterpret_cast<float *>(lmisc)[0] = __builtin_frexpf(raw, &k);
                            lmisc[1] = Q.e + k;
                        }
                    }
            }
            }
        }
        __syncthreads();  // (every wavefront's stores are out: the rows and their exponents are in L2)
        __builtin_amdgcn_s_dcache_inv();
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
            if (threadIdx.x < TRS_MAX_NW) prog[threadIdx.x] = 0x7fffffff;  // now: the LOWEST row whose neighbour cell is out
            __syncthreads();
            int smax = -(1 << 30);
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
                const int X0 = st.X;
                Bases<R> bx, by;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    bx.b[r] = base8<RS_XS>(E.X, lX, X0 + jr[r]);
                    by.b[r] = base8(E.Y, lY, (st.dl + 1) - X0 - jr[r]);  // as of anti-diagonal dl + 1
                }
                Feed fy;
                feed8_init<-1>(fy, E.Y, lY, st.dl - X0 - (K - 1), lane);
                TrsState Q;
                Q.A = zero_rdiag<R>(), Q.B = zero_rdiag<R>(), Q.carry = zero_rcell(), Q.e = TRS_E_NONE;
                const uint64_t out_lane = 1ull;  // lane 0 holds the stripe's first column in its register 0
                int blk_lo = 0, blk_hi = 0;
                RFRow<R> fa, fb;
#pragma unroll
                for (int r = 0; r < R; ++r) fa.v[r] = fb.v[r] = 0.f;
                cptr32 rm = rowmask + st.row0 + static_cast<uint32_t>(st.dl - st.df);
                const __amdgpu_buffer_rsrc_t rsF = stripe_rsrc(F, st.row0, K * 8), rsE = stripe_rsrc(Eb, st.row0, 4 * TRS_EDGE);
                // the forward rows' exponents: word 5 of the forward neighbour record of the same row
                cptr_i32 fexp = (cptr_i32)(Ef + static_cast<int64_t>(st.row0) * (4 * TRS_EDGE) + 20);
                Masks<R> mk_n = row_masks(rm[0]);
                trs_load_row(rsF, voff + (st.dl - st.df) * (K * 8), fb, mk_n);
                int ef_n = fexp[static_cast<int64_t>(st.dl - st.df) * TRS_EDGE];  // exponent of the row in hand, one step ahead like its row
                blk_lo = lenR, blk_hi = lenR;
                {   // (x+1, y+1) of the top slot on the first anti-diagonal: the right stripe's cell on dl + 2
                    const int q0 = st.dl + 2 - dfR;
                    if (q0 >= 0 && q0 < lenR) {
                        const int lo = max(q0 - TRS_BLOCK + 1, 0);
                        const int need = static_cast<int>(row0R) + lo;
                        while (uni(lds_peek(prog + wR)) > need) __builtin_amdgcn_s_sleep(2);
                        asm volatile("" ::: "memory");
                        trs_edge_stage(Eb, row0R + lo, q0 - lo + 1, stage, lane);
                        blk_lo = lo, blk_hi = q0 + 1;
                        const RCell c0 = trs_take_edge(Q, trs_edge_get(stage, q0 - lo));
                        if (lane == WAVE - 1) Q.carry = c0;
                    }
                }

                // f: the forward row of d (loaded a step ago); fnext: where the row of d-1 goes
                auto step = [&](int d, RDiag<R> &io, const RDiag<R> &s1, RFRow<R> &f, RFRow<R> &fnext) {
                    const int k = d - st.df;
                    const Masks<R> mk = mk_n;
                    const int ef = ef_n;
                    if (d > st.df) {
                        rm -= 1;
                        mk_n = row_masks(rm[0]);
                        trs_load_row(rsF, voff + (k - 1) * (K * 8), fnext, mk_n);
                        ef_n = fexp[static_cast<int64_t>(k - 1) * TRS_EDGE];
                    }
                    RCell edge = zero_rcell();
                    const int q = d + 1 - dfR;
                    if (static_cast<unsigned>(q) < static_cast<unsigned>(lenR)) {  // uniform
                        if (q < blk_lo) {
                            const int lo = max(q - TRS_BLOCK + 1, 0);
                            const int need = static_cast<int>(row0R) + lo;
                            while (uni(lds_peek(prog + wR)) > need) __builtin_amdgcn_s_sleep(2);
                            asm volatile("" ::: "memory");
                            trs_edge_stage(Eb, row0R + lo, q - lo + 1, stage, lane);
                            blk_lo = lo, blk_hi = q + 1;
                        }
                        edge = trs_take_edge(Q, trs_edge_get(stage, q - blk_lo));
                    }
                    bases_up<R>(by, feed8_get<-1>(fy, E.Y, lY, d - X0 - (K - 1), lane));
                    // one backward anti-diagonal: io d+2 -> d; s1 d+1; carry: the slot-above copy of d+2's register 0 -> that of d+1
                    const RCell Xe = dpp_rcell_from_above(s1.c[0], edge);  // (x+1, y) of every lane's top register
                    RDiag<R> o;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float em, exs, exl, eys, eyl;
                        rs_emissions(E.ltab, lanes_of(mk.cell[r]) ? bx.b[r] : RS_DEADX, lanes_of(mk.cell[r]) ? by.b[r] : RS_DEAD8, em, exs, exl, eys, eyl);
                        o.c[r] = rs_bwd_cell(E.tr, r + 1 < R ? io.c[r + 1] : Q.carry, r + 1 < R ? s1.c[r + 1] : Xe, s1.c[r], em, exs, exl, eys, eyl);
                    }
                    io = o;
                    Q.carry = Xe;
                    if (d == D) {  // the end corner (lX, lY)
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (X0 + jr[r] == lX) {
                                RCell c;
                                c.m = mdl->end[re * 5 + 0], c.sx = mdl->end[re * 5 + 1], c.sy = mdl->end[re * 5 + 2];
                                c.lx = mdl->end[re * 5 + 3], c.ly = mdl->end[re * 5 + 4];
                                io.c[r] = c;
                            }
                        Q.e = 0;
                    }
                    if ((d & (RS_K - 1)) == 0 && d < D) trs_renorm(Q);
                    trs_edge_store(rsE, k, io.c[0], Q.e, out_lane);
                    // posteriors of this anti-diagonal, slots claimed from the workgroup's LDS counter
                    {
                        const int sx = ef + Q.e - tot_e;
                        smax = max(smax, Q.e == TRS_E_NONE ? -(1 << 30) : sx);
                        float p[R];
                        uint64_t hit[R];
                        int total = 0;
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            p[r] = rs_posterior(f.v[r], io.c[r].m, sx, inv_tot);
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
                                    if (lanes_of(hit[r]) && slot < sink.cap) {
                                        sink.px[sink.off + slot] = X0 + jr[r] - 1 + sink.xs;
                                        sink.py[sink.off + slot] = y0 - jr[r] - 1 + sink.ys;
                                        sink.pp[sink.off + slot] = p[r];
                                    }
                                    base += __popcll(hit[r]);
                                }
                            }
                        }
                    }
                    if (((st.dl - d) & (TRS_BLOCK - 1)) == TRS_BLOCK - 1 || d == st.df) {
                        wait_vm();
                        if (lane == 0) lds_poke(prog + wv, static_cast<int>(st.row0) + k);
                    }
                };
                int d = st.dl;
                for (; d - 1 >= st.df; d -= 2) {
                    step(d, Q.B, Q.A, fb, fa);
                    step(d - 1, Q.A, Q.B, fa, fb);
                }
                if (d >= st.df) step(d, Q.B, Q.A, fb, fa);
                if (s == 0) {  // total from the backward side: the lattice point (0, 0) is the stripe's first slot on d = 0
                    const bool inB = ((st.dl - st.df) & 1) == 0;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (X0 + jr[r] == 0) {
                            const RCell cz = inB ? Q.B.c[r] : Q.A.c[r];
                            const float raw = rs_dot5(mdl->start + rs * 5, cz);
                            if (raw > 0.f) {
                                int k;
                                reinterpret_cast<float *>(lmisc)[2] = __builtin_frexpf(raw, &k);
                                lmisc[3] = Q.e + k;
                            }
                        }
                }
                }
            }
            if (lane == 0) atomicMax(lmisc + 6, smax);
            __syncthreads();
            out.btot_m = unif(reinterpret_cast<float *>(lmisc)[2]);
            out.btot_e = uni(lmisc[3]);
        }
        if (threadIdx.x == 0) {
            const int cnt = lmisc[4];
            out.npairs = cnt;
            if (cnt > pair_cap) out.status = NPR_ERR_CAPACITY;
            if (!alive || lmisc[6] >= TRS_S_LIMIT) out.status = TASK_RERUN;  // one exponent per stripe row may not have been enough (or nothing arrived: k_dp_tile decides)
            a.outs[t] = out;
            lmisc[5] = atomicAdd(a.queue, 1);
        }
        __syncthreads();
        t = uni(lmisc[5]) + static_cast<int>(gridDim.x);
    }
}

}  // namespace

size_t tile_rs_lds_bytes(int nw) {
    return sizeof(float) * (((RS_TABLE_FLOATS + 3) & ~3) + MODEL_FLOATS + 8 + TRS_MAX_NW + static_cast<size_t>(nw) * TRS_BLOCK * TRS_EDGE);
}

int launch_tile_rs(const KernelArgs &a, int NW, int grid, void *stream) {
    if (NW < 1 || NW > TRS_MAX_NW) return static_cast<int>(hipErrorInvalidValue);
    hipLaunchKernelGGL(k_dp_tile_rs, dim3(grid), dim3(WAVE * NW), tile_rs_lds_bytes(NW), static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}

}  // namespace npr
