// npr_kernel_tile_cs.hip -- k_dp_tile_cs<SW, FLAT>: the column-stripe kernel for WIDE bands (k_dp_tile's mapping, npr_kernel_tile.hip)
// in COLUMN-SCALED arithmetic: one binary exponent per LANE of a stripe -- per pair of lattice columns -- instead of one per cell
// (npr_cell.h: ~20 of a cell's ~55 vector instructions per sweep) or one per anti-diagonal row (npr_rs.h: cheap, but a row of a
// 3000-cell-wide rectangle spans more binary orders than fp32 has).
//
// Same recurrences -- cactus_realign's banded five-state forward / backward / posterior pass, SURVEY.md 8a rows a5.3-a5.5, reference
// call site nanopore/analyses/utils.py:587, for the band the reference's own parameters give (anchors +- diagonalExpansion 10, 14
// trimmed columns, splitMatrixBiggerThanThis 3000) --, same stripes, stripe tables, row masks and barrier-free pipeline of wavefronts
// as k_dp_tile, same outputs.  What differs:
//   * slot j of a stripe is lattice column X + j for the stripe's whole life, so a LANE sees one pair of columns: its values change
//     by a few binary orders per step, never by hundreds.  A lane keeps five plain fp32 values per cell relative to 2^e, e its own
//     (a VGPR); the recurrence inside a lane is npr_rs.h's 16-20 multiplies and FMAs and nothing else;
//   * what a lane takes from its neighbour (the lane below in the forward sweep, above in the backward sweep: one DPP move per state
//     and step, as in k_dp_tile) is multiplied by c = 2^(e_neighbour - e), a per-lane constant between two renormalisations;
//   * after every 16th anti-diagonal (d % 16 == 0 forward, 15 backward: the rows [16k, 16k+15] share their exponents in both sweeps)
//     every lane brings its largest value to 2^TCS_TOP and the exponents are made LIPSCHITZ along the data flow: e_l >= e_(l-1) -
//     TCS_C (one prefix maximum across the wavefront), so that c never exceeds 2^TCS_C and nothing overflows however far a value
//     travels inside a block (16 steps = 8 lanes: 2^(TCS_TOP + 6 + 8 TCS_C) < 2^127);
//   * the cell a stripe hands to its neighbour stripe travels with its lane's exponent, constant over a block of 16 rows: the blocks
//     of neighbour cells are staged aligned to those rows and the receiver's lane 0 (63) treats the exponent like a neighbour lane's;
//   * slots outside the band take the dead base code: all emissions 0, every state an exact zero (npr_rs.h) -- no EXEC masks;
//   * a forward row is 4 bytes per cell in the scratch, written whole; the lanes' exponents of a block go into the unused half of the
//     block's first row;
//   * the range certificate is per lane: a value a sweep flushed (below 2^-126 in its lane's units) times the largest value the other
//     sweep can hold in that lane and block (its exponent maximised over the lanes a value can have come from) must stay below
//     2^-89 of the total; a task with a lane above that (TCS_S_LIMIT) runs again in k_dp_tile (TASK_RERUN, npr_device.h).
// Scaling by powers of two is exact: wherever nothing leaves fp32's range relative to its lane the results are those of the
// per-cell-exponent kernels bit for bit, and the parity tests compare them with that mirror (tests/test_gpu_tile.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <cstdlib>

#include "npr_device.h"
#include "npr_frame.h"
#include "npr_rs.h"

namespace npr {

namespace {

constexpr int TCS_MAX_NW = 8;           // wavefronts per workgroup (launch bound)
constexpr int TCS_BLOCK = 16;           // rows that share their exponents = neighbour cells staged / published at a time
static_assert(TCS_BLOCK == RS_K && TCS_BLOCK == 16, "the blocks of k_dp_tile_cs are npr_rs.h's");
constexpr int TCS_EDGE = 8;             // one neighbour cell in memory: m, sx, sy, lx | ly, e, e^, -
constexpr int TCS_TOP = 50;             // a renormalised lane's largest value lies in [2^49, 2^50)
constexpr int TCS_C = 8;                // a lane's exponent is at least (the exponent of the lane its data comes from) - TCS_C
constexpr int TCS_REACH = 8;            // lanes a value can cross inside one block (16 steps, two slots per lane)
constexpr int TCS_NONE = -(1 << 24);    // exponent of "nothing there"
constexpr int TCS_BIAS = 1 << 25;       // makes every exponent a positive number for the unsigned wave scan
// lost mass per flushed value < 2^(cert + TCS_TOP + 6 - 126 + 1) of the total, cert = e^F + e^B - eTot of its lane and block; at most 2^28
// values (cells x states x sweeps) per task: below 2^-60 in all while cert stays below this
constexpr int TCS_S_LIMIT = 126 - 60 - 28 - (TCS_TOP + 6) - 1;
static_assert(TCS_TOP + 6 + TCS_REACH * TCS_C < 127, "a value that crosses TCS_REACH lanes inside a block must stay finite");

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

// One row per anti-diagonal of a stripe: 128 cells of 4 bytes, lane l at 8 l, in the first half of the row's space (the region is laid
// out for k_dp_tile's 8-byte cells, which may have to run the task again); bytes 512 + 8 l of a block's first row: (e, e^) of lane l.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stripe_rsrc(char *base, uint32_t row0, int row_bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base + static_cast<int64_t>(row0) * row_bytes, 0, -1, 0x00020000);
}
constexpr int TCS_ROW_BYTES = 1024, TCS_EXP_AT = 512;

// `cnt` neighbour cells of the rows [first, first + 16) of the neighbour stripe into this wavefront's LDS staging, record i = row first + i;
// rows outside [lo_row, hi_row] (the neighbour stripe has none there) become zeros.  Lane i takes record i; the loads bypass the vector
// L1 (sc1): the producer is another wavefront of this workgroup.  Returns the records' (e, e^) in the lanes that loaded one.
struct EdgeExp {
    int e, eh;
};
__device__ __forceinline__ EdgeExp tcs_edge_stage(char *Eb, uint32_t row0N, int dfN, int first, int lo_row, int hi_row, float *stage, int lane) {
    const int row = first + lane;
    v4i q = v4i{0, 0, 0, 0}, g = v4i{0, TCS_NONE, TCS_NONE, 0};
    if (lane < TCS_BLOCK) {
        if (row >= lo_row && row <= hi_row) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Eb + (static_cast<int64_t>(row0N) - dfN) * (4 * TCS_EDGE), 0, -1, 0x00020000);
            q = __builtin_amdgcn_raw_buffer_load_b128(rs, 32 * row, 0, 16);
            g = __builtin_amdgcn_raw_buffer_load_b128(rs, 32 * row + 16, 0, 16);
        }
        *reinterpret_cast<v4i *>(stage + TCS_EDGE * lane) = q;
        *reinterpret_cast<v4i *>(stage + TCS_EDGE * lane + 4) = g;
    }
    return EdgeExp{g.y, g.z};
}
__device__ __forceinline__ RCell tcs_edge_get(const float *stage, int i) {
    const float4 q = *reinterpret_cast<const float4 *>(stage + TCS_EDGE * i);
    const float g = stage[TCS_EDGE * i + 4];
    return RCell{q.x, q.y, q.z, q.w, g};
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

// 2^k, k clamped to [-127, 127]; 2^-127 and below: 0 (what is that far down is flushed)
__device__ __forceinline__ float tcs_pow2(int k) { return bitsf((min(max(k, -127), 127) + 127) << 23); }

// inclusive prefix maximum across the wavefront (lane l: the largest of lanes 0 .. l), positive 32-bit numbers: npr_rs.h's wave_max_u32 without
// the final read of lane 63
__device__ __forceinline__ uint32_t tcs_prefix_max(uint32_t v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 0"
                 : "+v"(v));
    return v;
}
__device__ __forceinline__ int tcs_lane_get(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(4 * src_lane, v); }

// What one lattice column costs: the largest transition x emission of a move that advances x (into match, shortGapX, longGapX, from any state) is
// at most 2^-(this many) -- rounded towards 0, clamped to [0, 8].  Both sweeps advance columns by the same moves.
__device__ __forceinline__ int tcs_column_cost(const DevModel *m) {
    float g = 0.f;
    const int to[3] = {0, 1, 3};
    for (int j = 0; j < 3; ++j) {
        float tmax = 0.f, emax = 0.f;
        for (int from = 0; from < 5; ++from) tmax = fmaxf(tmax, m->T[from * 5 + to[j]]);
        if (j == 0)
            for (int i = 0; i < 25; ++i) emax = fmaxf(emax, m->em[i]);
        else
            for (int i = 0; i < 5; ++i) emax = fmaxf(emax, m->ex[to[j] * 5 + i]);
        g = fmaxf(g, tmax * emax);
    }
    if (!(g > 0.f)) return 8;
    int k;
    (void)__builtin_frexpf(g, &k);  // g = f 2^k, f in [0.5, 1): -log2 g >= -k
    return min(max(-k, 0), 8);
}

// A stripe's register state: the rows of the two previous anti-diagonals and the carried neighbour copy in THIS lane's units 2^e; c = 2^(e of
// the lane the data comes from - e); eh: the largest e of the lanes a value of this block can have come from (the certificate's bound).
// carry: what slot 0 (forward) / the top slot (backward) takes from the neighbour lane's cell of TWO anti-diagonals ago -- forward the dot product of
// that cell's states with the transitions into match (made by the lane that owns the cell: umA / umB hold it for the top cells of A / B), backward the
// cell's match value -- already in this lane's units.  Of the neighbour's cell of ONE anti-diagonal ago a step takes m, sx, lx (and sy under SW) as they
// are, in the neighbour's units: c is folded into the x-gap emissions that multiply them (a power of two: the same bits).
struct CsState {
    RDiag<2> A, B;
    float carry, umA, umB;
    int e, eh;
    float c;
};
// The exponents of a stripe nothing has entered yet: going down by TCS_C per lane from where its first values will appear -- from the neighbour
// stripe's edge lane (origin -1 / 64, exponent e_in; TCS_NONE: no neighbour, 0) or from the lane of the start / end cell (whose exponent is then -TCS_TOP).
template <bool FWD>
__device__ __forceinline__ void tcs_init(CsState &Q, int lane, int e_in, int eh_in, int origin, int dslot) {
    Q.A = zero_rdiag<2>(), Q.B = zero_rdiag<2>(), Q.carry = 0.f, Q.umA = 0.f, Q.umB = 0.f;
    const bool cell = origin >= 0 && origin < WAVE;  // the start / end cell goes in at 2^TCS_TOP like a renormalised value
    const int base = cell ? -TCS_TOP : (e_in == TCS_NONE ? 0 : e_in);
    const int src = FWD ? lane - 1 : lane + 1;  // the lane this one's neighbour values come from
    Q.e = base - TCS_C * abs(lane - origin);
    // what can be in this lane before the first renormalisation: what the neighbour stripe's edge lane holds, TCS_REACH lanes far at most and discounted
    // by the columns in between (tcs_renorm); around the start / end cell: that cell's 2^0
    const int pos = FWD ? lane : WAVE - 1 - lane;
    Q.eh = cell ? base : (pos <= TCS_REACH ? eh_in - dslot * (2 * pos + 1) : TCS_NONE);
    Q.c = (src < 0 || src >= WAVE) ? (e_in == TCS_NONE ? 0.f : tcs_pow2(e_in - Q.e)) : tcs_pow2(TCS_C * (abs(lane - origin) - abs(src - origin)));
}
// Renormalisation of everything a stripe holds, lane by lane, with the exponents kept Lipschitz along the data flow; e_in / eh_in: the
// neighbour stripe's edge lane for the block to come (uniform; TCS_NONE: it has no rows there).
// Returns the biased exponent of the largest value the lane held (0: nothing).
template <bool FWD>
__device__ __forceinline__ int tcs_renorm(CsState &Q, int lane, int e_in, int eh_in, int dslot) {
    const int pos = FWD ? lane : WAVE - 1 - lane;
    uint32_t u = static_cast<uint32_t>(fbits(Q.carry));
#pragma unroll
    for (int r = 0; r < 2; ++r) u = umax3(u, rcell_max_bits(Q.A.c[r]), rcell_max_bits(Q.B.c[r]));
    const int eb = static_cast<int>(u >> 23);  // biased exponent of the lane's largest value; 0: nothing (or less than a normal number)
    const int own = eb ? Q.e + eb - (126 + TCS_TOP) : TCS_NONE;
    int t = own + TCS_C * pos + TCS_BIAS;
    if (pos == 0) t = max(t, e_in - TCS_C + TCS_BIAS);
    if constexpr (!FWD) t = tcs_lane_get(t, WAVE - 1 - lane);
    t = static_cast<int>(tcs_prefix_max(static_cast<uint32_t>(t)));
    if constexpr (!FWD) t = tcs_lane_get(t, WAVE - 1 - lane);
    const int en = t - TCS_BIAS - TCS_C * pos;
    const float f = tcs_pow2(Q.e - en);
#pragma unroll
    for (int r = 0; r < 2; ++r) rcell_scale(Q.A.c[r], f), rcell_scale(Q.B.c[r], f);
    Q.carry *= f, Q.umA *= f, Q.umB *= f;
    const int eu = FWD ? dpp_from_below(en, e_in) : dpp_from_above(en, e_in);
    Q.c = tcs_pow2(eu - en);
    Q.e = en;
    // What bounds the values this lane can hold during the coming block: its own exponent, and the exponents of the TCS_REACH lanes a value can
    // arrive from, each DISCOUNTED by what the trip costs -- a value that advances one lattice column was multiplied by a transition and an
    // emission, at most 2^-dslot (tcs_column_cost), and from k lanes away it advances 2 k - 1 columns at least.  Without the discount the bound
    // counts the natural decay away from an alignment's ridge (4-5 binary orders per lane) as if it were headroom used: one task in five then
    // fails a certificate that nothing threatens.
    // A lane that holds nothing (own == TCS_NONE: its exponent is only its place in the chain) has nothing to send and bounds nothing.
    const int have = own == TCS_NONE ? TCS_NONE : en;
    int w = tcs_lane_get(have, FWD ? max(lane - 1, 0) : min(lane + 1, WAVE - 1)) - dslot;  // k = 1 (the edge lane reads itself: harmless, it is below `have`)
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // k = 2 .. 8: windows of 2, 4, 8 lanes, 2 dslot per lane
        const int s = 1 << i;
        w = max(w, tcs_lane_get(w, FWD ? max(lane - s, 0) : min(lane + s, WAVE - 1)) - 2 * dslot * s);
    }
    const int m = max(have, w);
    Q.eh = pos <= TCS_REACH ? max(m, eh_in - dslot * (2 * pos + 1)) : m;
    return eb;
}

__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Waiting for a neighbour stripe's progress word.  Bounded: a table that promised rows no wavefront will ever write must not hang the device --
// after ~2^22 polls (seconds) the wait gives up and the task is reported for the second pass (TASK_RERUN), whose kernel has the same bound.
constexpr int TCS_SPIN_LIMIT = 1 << 22;
__device__ __forceinline__ void tcs_wait_at_least(const int *p, int need, int &stuck) {
    int spins = 0;
    while (uni(lds_peek(p)) < need && spins < TCS_SPIN_LIMIT) __builtin_amdgcn_s_sleep(2), ++spins;
    if (spins >= TCS_SPIN_LIMIT) stuck = 1;
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void tcs_wait_at_most(const int *p, int need, int &stuck) {
    int spins = 0;
    while (uni(lds_peek(p)) > need && spins < TCS_SPIN_LIMIT) __builtin_amdgcn_s_sleep(2), ++spins;
    if (spins >= TCS_SPIN_LIMIT) stuck = 1;
    asm volatile("" ::: "memory");
}

// ---- one anti-diagonal of a stripe, the part every step has ----
__device__ __forceinline__ float tcs_dpp_below(float v, float edge) { return bitsf(__builtin_amdgcn_update_dpp(fbits(edge), fbits(v), 0x138, 0xf, 0xf, false)); }
__device__ __forceinline__ float tcs_dpp_above(float v, float edge) { return bitsf(__builtin_amdgcn_update_dpp(fbits(edge), fbits(v), 0x130, 0xf, 0xf, false)); }
// the transitions into match applied to a cell: what the cell diagonally above-right of it starts from (rs_fwd_cell's first five operations)
__device__ __forceinline__ float tcs_into_match(const Trans &t, const RCell &c) {
    float a = t.mm * c.m;
    a = __builtin_fmaf(t.sxm, c.sx, a);
    a = __builtin_fmaf(t.sym, c.sy, a);
    a = __builtin_fmaf(t.lxm, c.lx, a);
    a = __builtin_fmaf(t.lym, c.ly, a);
    return a;
}
// forward: io d-2 -> d; p1 d-1; um_p1: tcs_into_match of p1's top cell, um_io: that of io's new one; erec: the left stripe's record of d-1 (LDS: um, m, sx,
// lx | sy).  Same operations in the same order as rs_fwd_cell on every cell (npr_rs.h); only WHO makes the match dot product of a lane's top cell differs.
// FULL: every slot of the row is a band cell (its mask word is 0): the emissions come without the in-band selects
template <bool SW, bool FLAT, bool FULL = false>
__device__ __forceinline__ void tcs_fwd_core(const StepEnv &E, RDiag<2> &io, const RDiag<2> &p1, float &carry, float &um_io, float um_p1, float c, const Masks<2> &mk,
                                             const float *erec, const Bases<2> &bx, Bases<2> &by, int inject) {
    const float4 edge = *reinterpret_cast<const float4 *>(erec);
    bases_down<2>(by, inject);
    const float um_s = tcs_dpp_below(um_p1, edge.x) * c;  // (x-1, y-1)'s share of the NEXT anti-diagonal's slot 0, in this lane's units from here on
    const float Lm = tcs_dpp_below(p1.c[1].m, edge.y), Lsx = tcs_dpp_below(p1.c[1].sx, edge.z), Llx = tcs_dpp_below(p1.c[1].lx, edge.w);  // (x-1, y), the neighbour's units
    float Lsy = 0.f;
    if constexpr (SW) Lsy = tcs_dpp_below(p1.c[1].sy, erec[4]);
    float em[2], exs[2], exl[2], eys[2], eyl[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) rs_cell_emissions<FULL ? 4 : 2, FLAT>(E.ltab, mk.cell[r], bx.b[r], by.b[r], em[r], exs[r], exl[r], eys[r], eyl[r]);
    RDiag<2> o;
    {
        const Trans &t = E.tr;
        const RCell &U = p1.c[0];
        const float exs_c = exs[0] * c, exl_c = FLAT ? exs_c : exl[0] * c;
        RCell q;
        float a;
        q.m = em[0] * carry;
        a = t.msx * Lm;
        a = __builtin_fmaf(t.sxsx, Lsx, a);
        if constexpr (SW) a = __builtin_fmaf(t.sysx, Lsy, a);
        q.sx = exs_c * a;
        a = t.mlx * Lm;
        a = __builtin_fmaf(t.lxlx, Llx, a);
        q.lx = exl_c * a;
        a = t.msy * U.m;
        a = __builtin_fmaf(t.sysy, U.sy, a);
        if constexpr (SW) a = __builtin_fmaf(t.sxsy, U.sx, a);
        q.sy = eys[0] * a;
        a = t.mly * U.m;
        a = __builtin_fmaf(t.lyly, U.ly, a);
        q.ly = eyl[0] * a;
        o.c[0] = q;
    }
    o.c[1] = rs_fwd_cell<SW>(E.tr, p1.c[0], io.c[0], p1.c[1], em[1], exs[1], exl[1], eys[1], eyl[1]);
    io = o;
    carry = um_s;
    um_io = tcs_into_match(E.tr, o.c[1]);
}
// backward: io d+2 -> d; s1 d+1; carry: the match value of (x+1, y+1) of the top slot, in this lane's units; erec: the right stripe's record of d+1 (m, sx, lx)
template <bool SW, bool FLAT, bool FULL = false>
__device__ __forceinline__ void tcs_bwd_core(const StepEnv &E, RDiag<2> &io, const RDiag<2> &s1, float &carry, float c, const Masks<2> &mk, const float *erec,
                                             const Bases<2> &bx, Bases<2> &by, int inject) {
    const float4 edge = *reinterpret_cast<const float4 *>(erec);
    bases_up<2>(by, inject);
    const float Xm = tcs_dpp_above(s1.c[0].m, edge.x) * c;  // becomes (x+1, y+1) of the next anti-diagonal
    const float Xsx = tcs_dpp_above(s1.c[0].sx, edge.y), Xlx = tcs_dpp_above(s1.c[0].lx, edge.z);  // (x+1, y), the neighbour's units
    float em[2], exs[2], exl[2], eys[2], eyl[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) rs_cell_emissions<FULL ? 4 : 2, FLAT>(E.ltab, mk.cell[r], bx.b[r], by.b[r], em[r], exs[r], exl[r], eys[r], eyl[r]);
    RDiag<2> o;
    o.c[0] = rs_bwd_cell<SW>(E.tr, io.c[1], s1.c[1], s1.c[0], em[0], exs[0], exl[0], eys[0], eyl[0]);
    {
        const Trans &t = E.tr;
        const RCell &Ys = s1.c[1];
        const float exs_c = exs[1] * c, exl_c = FLAT ? exs_c : exl[1] * c;
        const float am = em[1] * carry, asx = exs_c * Xsx, alx = exl_c * Xlx, asy = eys[1] * Ys.sy, aly = eyl[1] * Ys.ly;
        RCell q;
        float b;
        b = t.mm * am;
        b = __builtin_fmaf(t.msx, asx, b);
        b = __builtin_fmaf(t.mlx, alx, b);
        b = __builtin_fmaf(t.msy, asy, b);
        b = __builtin_fmaf(t.mly, aly, b);
        q.m = b;
        b = t.sxm * am;
        b = __builtin_fmaf(t.sxsx, asx, b);
        if constexpr (SW) b = __builtin_fmaf(t.sxsy, asy, b);
        q.sx = b;
        b = t.sym * am;
        b = __builtin_fmaf(t.sysy, asy, b);
        if constexpr (SW) b = __builtin_fmaf(t.sysx, asx, b);
        q.sy = b;
        b = t.lxm * am;
        b = __builtin_fmaf(t.lxlx, alx, b);
        q.lx = b;
        b = t.lym * am;
        b = __builtin_fmaf(t.lyly, aly, b);
        q.ly = b;
        o.c[1] = q;
    }
    io = o;
    carry = Xm;
}
__device__ __forceinline__ void tcs_swap(float &x, float &y) {
    const float t = x;
    x = y, y = t;
}
// the two held rows trade places (a generic step is written for an odd anti-diagonal: the row it makes goes to Q.A)
__device__ __forceinline__ void tcs_swap_rows(RDiag<2> &A, RDiag<2> &B) {
    const RDiag<2> t = A;
    A = B, B = t;
}
template <int AUX = 0>
__device__ __forceinline__ void tcs_store_row(__amdgpu_buffer_rsrc_t rsF, int vo, const RDiag<2> &io) {
    __builtin_amdgcn_raw_buffer_store_b64(v2i{fbits(io.c[0].m), fbits(io.c[1].m)}, rsF, vo, 0, AUX);
}
// the record a stripe leaves for its neighbour stripe: forward (um, m, sx, lx | sy, e, e^, ly) of its last column's cell, backward (m, sx, lx, - | -, e, e^, -)
// of its first column's
__device__ __forceinline__ void tcs_store_edge_fwd(__amdgpu_buffer_rsrc_t rsE, int vo, const RCell &c, float um, int e, int eh) {
    __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(um), fbits(c.m), fbits(c.sx), fbits(c.lx)}, rsE, vo, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(c.sy), e, eh, fbits(c.ly)}, rsE, vo + 16, 0, 0);
}
// E-step: the other four states of a row, 12 bytes per cell in the planes' region: each as the top 24 bits of its fp32 word (the sign, the exponent
// whole, 15 mantissa bits, rounded), four to three words.  The forward sweep of the E-step is bound by the bytes it stores (13 ms of a 34 ms sweep
// without these planes; as 16-byte cells they cost 21 ms), and what a count needs of a forward value is its leading bits -- the counts add thousands
// of them in fp32; half floats (8 bytes) lack the range: a lane's long-gap states lie 2^-10 .. 2^-20 below its match values.
// Row stride 1536: lane l's words 0-3 (slot 0's sx sy lx ly and the first byte-triple of slot 1's) at 16 l, words 4-5 at 1024 + 8 l.
constexpr int TCS_XROW_BYTES = 1536;
// The E-step's rows are written once and read once, tens of milliseconds later, by a sweep that moves 39 bytes per cell at 2.9 TB/s: as non-temporal
// stores and loads (aux 2: nt) they stay out of the way of what the caches can help with -- the records between stripes, the exponents, the tables:
// 55.2 -> 51.0 ms per step of the bench's batch (four runs, alternating: 55.2 / 56.3 against 51.0 / 50.7); nt on the planes alone 51.5, sc0 | nt 51.2.
#ifndef NPR_TCS_EM_AUX
#define NPR_TCS_EM_AUX 2
#endif
#ifndef NPR_TCS_EM_ROW_AUX
#define NPR_TCS_EM_ROW_AUX NPR_TCS_EM_AUX
#endif
constexpr int TCS_EM_ROW_AUX = NPR_TCS_EM_ROW_AUX;
constexpr int TCS_EM_AUX = NPR_TCS_EM_AUX;  // cache policy of the planes' stores and of the backward sweep's row loads
__device__ __forceinline__ void tcs_pack24(const RCell &c, int &d0, int &d1, int &d2) {
    const uint32_t r0 = static_cast<uint32_t>(fbits(c.sx)) + 0x80u, r1 = static_cast<uint32_t>(fbits(c.sy)) + 0x80u, r2 = static_cast<uint32_t>(fbits(c.lx)) + 0x80u,
                   r3 = static_cast<uint32_t>(fbits(c.ly)) + 0x80u;  // (values are finite and not negative: the carry can only reach the exponent)
    d0 = static_cast<int>(__builtin_amdgcn_perm(r1, r0, 0x05030201u));  // r0's bytes 1 2 3, r1's byte 1
    d1 = static_cast<int>(__builtin_amdgcn_perm(r2, r1, 0x06050302u));  // r1's bytes 2 3, r2's bytes 1 2
    d2 = static_cast<int>(__builtin_amdgcn_perm(r3, r2, 0x07060503u));  // r2's byte 3, r3's bytes 1 2 3
}
__device__ __forceinline__ void tcs_unpack24(int d0, int d1, int d2, float &sx, float &sy, float &lx, float &ly) {
    const uint32_t u0 = static_cast<uint32_t>(d0), u1 = static_cast<uint32_t>(d1), u2 = static_cast<uint32_t>(d2);
    sx = bitsf(static_cast<int>(__builtin_amdgcn_perm(0u, u0, 0x0201000cu)));
    sy = bitsf(static_cast<int>(__builtin_amdgcn_perm(u1, u0, 0x0504030cu)));
    lx = bitsf(static_cast<int>(__builtin_amdgcn_perm(u2, u1, 0x0403020cu)));
    ly = bitsf(static_cast<int>(u2 & 0xffffff00u));
}
// (NPR_TCS_EM_EXP, bring-up builds only -- WRONG counts, for timing: 1 the E-step stops after its forward sweep, 2 its backward sweep loads no forward
// rows, 3 the forward sweep alone and without these planes.  How DESIGN.md 5.3e's "measured by leaving things out" numbers were taken.)
__device__ __forceinline__ void tcs_store_planes(__amdgpu_buffer_rsrc_t rsX, int vo_a, int vo_b, const RDiag<2> &io) {
#ifdef NPR_TCS_EM_EXP
    if (NPR_TCS_EM_EXP == 3) return;
#endif
    int d[6];
    tcs_pack24(io.c[0], d[0], d[1], d[2]);
    tcs_pack24(io.c[1], d[3], d[4], d[5]);
    __builtin_amdgcn_raw_buffer_store_b128(v4i{d[0], d[1], d[2], d[3]}, rsX, vo_a, 0, TCS_EM_AUX);
    __builtin_amdgcn_raw_buffer_store_b64(v2i{d[4], d[5]}, rsX, vo_b, 0, TCS_EM_AUX);
}
__device__ __forceinline__ void tcs_store_edge_bwd(__amdgpu_buffer_rsrc_t rsE, int vo, const RCell &c, int e, int eh) {
    __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(c.m), fbits(c.sx), fbits(c.lx), 0}, rsE, vo, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(v4i{0, e, eh, 0}, rsE, vo + 16, 0, 0);
}

// LDS of a workgroup (static: the table offsets fold into the LDS instructions, npr_rs.h)
template <int NWMAX>
struct __attribute__((aligned(16))) CsLds {
    float stage[NWMAX][TCS_BLOCK * TCS_EDGE];  // per wavefront: the neighbour stripe's block of cells
    RsTables tab;
    float model[MODEL_FLOATS];
    int misc[8];            // [0..3] totals, [4] pair counter, [5] next task, [6] largest certificate value, [7] column cost
    int prog[NWMAX];   // rows whose neighbour cells are out
};

// ... and what the E-step variant adds: per wavefront the emission bins (one column per lane, no atomics in the loop: k_em_tile's) and the block
// of the LEFT stripe's forward records its lane 0 counts transitions from
#ifndef NPR_TCS_EM_NW
#define NPR_TCS_EM_NW 1
#endif
constexpr int TCS_EM_NW = NPR_TCS_EM_NW;
constexpr int TCS_EM_ROWS = EM_BINS + 5;  // the bins' rows and a scratch zone for N bases: a short-gap bin's long-gap partner lies 4 rows on, there too
struct __attribute__((aligned(16))) CsEmLds {
    float stageF[TCS_EM_NW][TCS_BLOCK * TCS_EDGE];
    float bins[TCS_EM_NW][TCS_EM_ROWS * WAVE];
};
#ifndef NPR_TCS_EM_WAVES
#define NPR_TCS_EM_WAVES 2
#endif
// a lane whose backward values outgrew 2^(TCS_TOP + this) inside a block (values entering it down a steep exponent gradient) would count its
// transitions with forward factors scaled further down than fp32 holds exactly: the task is counted by k_em_tile instead
constexpr int TCS_EM_BOOST = 40;
#ifndef NPR_TCS_EM_DEAD
#define NPR_TCS_EM_DEAD (-40)
#endif
constexpr int TCS_EM_DEAD = NPR_TCS_EM_DEAD;  // (k_em_tile's EM_SKIP: 2.7e8 terms of 2^-40 each are 2.4e-4 of a count)
#ifndef NPR_TCS_WAVES
#define NPR_TCS_WAVES 6
#endif
#ifndef NPR_TCS_DP_MASK
#define NPR_TCS_DP_MASK 1
#endif
// (the DP instances' forward rows too are written once and read once, much later: as non-temporal stores and loads 270.9 / 270.0 -> 263.5 / 263.2 ms
// per launch of the reference's band, 8192 reads, alternating runs)
#ifndef NPR_TCS_DP_AUX
#define NPR_TCS_DP_AUX 2
#endif
constexpr int TCS_DP_AUX = NPR_TCS_DP_AUX;  // cache policy of the DP instances' row stores and loads (2: nt)
constexpr bool TCS_DP_MASK = NPR_TCS_DP_MASK != 0;  // the DP instances' row stores and loads in the fast loops only in the lanes that hold a band cell (what halved the E-step's traffic): with plain stores and loads 268.6 / 269.2 -> 272.9 / 273.1 ms per launch of the reference's band, with non-temporal ones 262.1 / 262.2 -> 260.0 / 259.7: on since then
#ifndef NPR_TCS_T_SGPR
#define NPR_TCS_T_SGPR 1
#endif
// EM: the Baum-Welch E-step on the same sweeps (k_em_tile's job, npr_kernel_tile.hip; nanopore/analyses/utils.py:509-528): the forward sweep also keeps
// the other four states of every cell, and the backward sweep, instead of emitting posteriors, adds the posterior of every transition into the cells of
// an anti-diagonal to 15 per-lane accumulators and of every emitted symbol to per-lane bins in LDS.  In this arithmetic a forward row comes back from
// memory scaled ONCE by 2^(eF + eB - eTot) of its lane and block, so a count is three multiplies and no exponent; cells outside the band are exact
// zeros on both sides, so there is no mask and no branch in the counting.
template <bool SW, bool FLAT, bool EM = false>
__global__ void __launch_bounds__(WAVE *(EM ? TCS_EM_NW : TCS_MAX_NW)) __attribute__((amdgpu_waves_per_eu(EM ? NPR_TCS_EM_WAVES : NPR_TCS_WAVES))) k_dp_tile_cs(KernelArgs a) {
    constexpr int NWMAX = EM ? TCS_EM_NW : TCS_MAX_NW;
    __shared__ CsLds<NWMAX> L;
    float *lbins = nullptr, *stageF = nullptr;
    if constexpr (EM) {
        __shared__ CsEmLds LE;
        lbins = LE.bins[uni(static_cast<int>(threadIdx.x) >> 6)], stageF = LE.stageF[uni(static_cast<int>(threadIdx.x) >> 6)];
    }
    float *lmodel = L.model;
    int *lmisc = L.misc;
    int *prog = L.prog;
    constexpr int R = 2, K = 64 * R;

    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = uni(static_cast<int>(threadIdx.x) >> 6);
    const int NW = static_cast<int>(blockDim.x) >> 6;
    float *const stage = L.stage[wv];
    char *const F = a.F + uni64(a.region[blockIdx.x]) * 8;
    char *const Fx = EM ? reinterpret_cast<char *>(a.Fx) + uni64(a.region[blockIdx.x]) * 16 : nullptr;  // E-step: the other four states, 16 bytes per cell
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
        char *const Ef = F + static_cast<int64_t>(rows) * (K * 8);          // neighbour cells of the forward sweep (+ their lane's exponents)
        char *const Eb = Ef + static_cast<int64_t>(rows) * (4 * TCS_EDGE);  // ... of the backward sweep
        const int rs = flags & 1, re = (flags >> 1) & 1;

        __syncthreads();
        {
            const float *gm = reinterpret_cast<const float *>(a.models + model);
            for (int i = threadIdx.x; i < MODEL_FLOATS; i += blockDim.x) lmodel[i] = gm[i];
            if (threadIdx.x == 0) lmisc[0] = 0, lmisc[1] = E_DEAD, lmisc[2] = 0, lmisc[3] = E_DEAD, lmisc[4] = 0, lmisc[6] = -(1 << 30);
            if (threadIdx.x < NWMAX) prog[threadIdx.x] = 0;
            if constexpr (EM)
                for (int i = 0; i < TCS_EM_ROWS; ++i) lbins[i * WAVE + lane] = 0.f;
        }
        __syncthreads();
        rs_build_tables(&L.tab, reinterpret_cast<const DevModel *>(lmodel), threadIdx.x, blockDim.x);
        if (threadIdx.x == 0) lmisc[7] = tcs_column_cost(reinterpret_cast<const DevModel *>(lmodel));
        __syncthreads();
        const int dslot = uni(lmisc[7]);
        StepEnv E;
        E.mdl = reinterpret_cast<const DevModel *>(lmodel);
        E.ltab = reinterpret_cast<const char *>(&L.tab);
        E.X = a.seq + x_off, E.Y = a.seq + y_off, E.lX = lX, E.lY = lY, E.lane = lane;
        {
            Trans tr = load_trans(E.mdl->T);
#if NPR_TCS_T_SGPR
            tr.mm = unif(tr.mm), tr.sxm = unif(tr.sxm), tr.sym = unif(tr.sym), tr.lxm = unif(tr.lxm), tr.lym = unif(tr.lym);
            tr.msx = unif(tr.msx), tr.sxsx = unif(tr.sxsx), tr.sysx = unif(tr.sysx);
            tr.msy = unif(tr.msy), tr.sysy = unif(tr.sysy), tr.sxsy = unif(tr.sxsy);
            tr.mlx = unif(tr.mlx), tr.lxlx = unif(tr.lxlx), tr.mly = unif(tr.mly), tr.lyly = unif(tr.lyly);
#endif
            E.tr = tr;
        }
        const DevModel *mdl = E.mdl;
        int stuck = 0;

        // =============================== forward ===============================
        // The row of an ODD anti-diagonal lives in Q.A, of an even one in Q.B (whatever the stripe's first row: both start as zeros).
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
            Bases<R> bx, by;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                bx.b[r] = base8<RS_XS>(E.X, lX, st.X + jr[r] - 1);
                by.b[r] = base8(E.Y, lY, (st.df - 1) - st.X - jr[r] - 1);  // as of anti-diagonal df - 1
            }
            Feed fy;
            feed8_init<+1>(fy, E.Y, lY, st.df - st.X - 1, lane);
            cptr32 rm = rowmask + st.row0;  // the word of the row in hand is rm[0]
            const __amdgpu_buffer_rsrc_t rsF = stripe_rsrc(F, st.row0, TCS_ROW_BYTES), rsE = stripe_rsrc(Ef, st.row0, 4 * TCS_EDGE);
            const __amdgpu_buffer_rsrc_t rsX = EM ? stripe_rsrc(Fx, st.row0, TCS_XROW_BYTES) : rsF;
            const bool edge_lane = lane == st.K / R - 1;  // holds the stripe's last column in its top register
            // the left stripe's block of rows [16 kb, 16 kb + 15]: wait until it is out, stage it, its lane-63 exponents
            auto take_block = [&](int kb, int &e_in, int &eh_in) {
                const int first = kb * TCS_BLOCK, lo = max(first, dfL), hi = min(first + TCS_BLOCK - 1, dlL);
                if (hi >= lo) {  // uniform
                    const int need = static_cast<int>(row0L) + (hi - dfL) + 1;
                    tcs_wait_at_least(prog + wL, need, stuck);
                }
                const EdgeExp x = tcs_edge_stage(Ef, row0L, dfL, first, lo, hi, stage, lane);
                e_in = hi >= lo ? __builtin_amdgcn_readlane(x.e, lo - first) : TCS_NONE;
                eh_in = hi >= lo ? __builtin_amdgcn_readlane(x.eh, lo - first) : TCS_NONE;
            };
            CsState Q;
            {
                int e_in, eh_in;
                take_block((st.df - 1) >> 4, e_in, eh_in);
                tcs_init<true>(Q, lane, e_in, eh_in, s == 0 ? 0 : -1, dslot);  // (the start cell (0, 0) is slot 0 of the first stripe)
                // (x-1, y-1) of slot 0 on the first anti-diagonal: the left stripe's cell on df - 2 -- in the block before when df - 1 opens one
                const int q0 = st.df - 2;
                if (q0 >= dfL && q0 <= dlL) {  // uniform
                    const int need = static_cast<int>(row0L) + (q0 - dfL) + 1;
                    tcs_wait_at_least(prog + wL, need, stuck);
                    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(Ef + (static_cast<int64_t>(row0L) + (q0 - dfL)) * (4 * TCS_EDGE), 0, -1, 0x00020000);
                    const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rq, 0, 0, 16);
                    const v4i g = __builtin_amdgcn_raw_buffer_load_b128(rq, 16, 0, 16);
                    if (lane == 0) Q.carry = __builtin_ldexpf(bitsf(q.x), g.y - Q.e);  // (the record's um, in this lane's units)
                }
                __builtin_amdgcn_raw_buffer_store_b64(v2i{Q.e, Q.eh}, rsF, TCS_EXP_AT + 8 * lane, 0, 0);
            }

            // any row: the start cell, the block boundaries, the hand-over
            auto step = [&](int d, RDiag<R> &io, const RDiag<R> &p1) {
                const int k = d - st.df;
                const Masks<R> mk = row_masks(rm[0]);
                rm += 1;
                tcs_fwd_core<SW, FLAT>(E, io, p1, Q.carry, Q.umA, Q.umB, Q.c, mk, stage + TCS_EDGE * ((d - 1) & (TCS_BLOCK - 1)), bx, by, feed8_get<+1>(fy, E.Y, lY, d - st.X - 1, lane));  // (io is Q.A, p1 is Q.B: gstep)
                if (d == 0) {  // the start cell (0, 0): slot 0 of the first stripe
                    if (lane == 0) {
                        const int k0 = -Q.e;
                        io.c[0] = RCell{__builtin_ldexpf(mdl->start[rs * 5 + 0], k0), __builtin_ldexpf(mdl->start[rs * 5 + 1], k0), __builtin_ldexpf(mdl->start[rs * 5 + 2], k0),
                                        __builtin_ldexpf(mdl->start[rs * 5 + 3], k0), __builtin_ldexpf(mdl->start[rs * 5 + 4], k0)};
                    }
                }
                if ((d & (TCS_BLOCK - 1)) == 0) {  // the rows from d on share new exponents; the left stripe's block of the steps to come
                    if (d < st.dl) {
                        int e_in, eh_in;
                        take_block(d >> 4, e_in, eh_in);
                        tcs_renorm<true>(Q, lane, e_in, eh_in, dslot);
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(v2i{Q.e, Q.eh}, rsF, k * TCS_ROW_BYTES + TCS_EXP_AT + 8 * lane, 0, 0);
                }
                if constexpr (EM) {  // (the lanes that hold a band cell only: the E-step is bound by these bytes)
                    if (lanes_of(mk.lanes)) {
                        tcs_store_row<TCS_EM_ROW_AUX>(rsF, voff + k * TCS_ROW_BYTES, io);
                        tcs_store_planes(rsX, 2 * voff + k * TCS_XROW_BYTES, 1024 + voff + k * TCS_XROW_BYTES, io);
                    }
                } else {
                    tcs_store_row<TCS_DP_AUX>(rsF, voff + k * TCS_ROW_BYTES, io);
                }
                if (edge_lane) tcs_store_edge_fwd(rsE, 4 * TCS_EDGE * k, io.c[R - 1], Q.umA, Q.e, Q.eh);
                if ((d & (TCS_BLOCK - 1)) == TCS_BLOCK - 1 || d == st.dl) {
                    wait_vm();
                    if (lane == 0) lds_poke(prog + wv, static_cast<int>(st.row0) + k + 1);
                }
            };
            // any row, whatever its parity: the step above is instantiated once, for an odd anti-diagonal
            auto gstep = [&](int d) {
                if (!(d & 1)) tcs_swap_rows(Q.A, Q.B), tcs_swap(Q.umA, Q.umB);
                step(d, Q.A, Q.B);
                if (!(d & 1)) tcs_swap_rows(Q.A, Q.B), tcs_swap(Q.umA, Q.umB);
            };
            // ONE loop with ONE instance of the general step (the kernel has to stay small for the instruction cache).  Where a row 16 kb + 1 opens a whole block
            // of sixteen rows inside the stripe, its first fifteen -- no corner cell, no boundary, one hand-over at the end: nothing but the recurrence, the
            // stores and the stream -- run in the fast loop (d is odd: pairs of an A step and a B step) and the general step takes the boundary row 16 kb + 16.
            int d = st.df;
            while (d <= st.dl) {
                if ((d & (TCS_BLOCK - 1)) == 1 && d + TCS_BLOCK - 1 <= st.dl) {
                feed8_ahead<+1>(fy, E.Y, lY, d - st.X - 1, lane);  // (the window serves the sixteen bases the block asks for)
                int yi = uni(d - st.X - 1 - fy.base);
                int vo = voff + (d - st.df) * TCS_ROW_BYTES, ve = 4 * TCS_EDGE * (d - st.df);
                [[maybe_unused]] int xo = (d - st.df) * TCS_XROW_BYTES;  // (E-step: the planes' row)
                const float *er = stage;  // record (d - 1) & 15 = 0
                // the rows' mask words by one vector load (lane i mod 16: row d + i), handed out by v_readlane: no scalar load to wait for in the loop
                const int wv16 = static_cast<int>(a.rowmask[rowmask_off + st.row0 + static_cast<uint32_t>(d - st.df) + (lane & (TCS_BLOCK - 1))]);
#pragma unroll 1
                for (int i = 0; i < 7; ++i) {
                    const Masks<R> m0 = row_masks(__builtin_amdgcn_readlane(wv16, 2 * i));
                    tcs_fwd_core<SW, FLAT>(E, Q.A, Q.B, Q.carry, Q.umA, Q.umB, Q.c, m0, er, bx, by, __builtin_amdgcn_readlane(fy.cur, yi));
                    if constexpr (EM) {
                        if (lanes_of(m0.lanes)) {
                            tcs_store_row<TCS_EM_ROW_AUX>(rsF, vo, Q.A);
                            tcs_store_planes(rsX, 2 * voff + xo, 1024 + voff + xo, Q.A);
                        }
                    } else {
                        if (!TCS_DP_MASK || lanes_of(m0.lanes)) tcs_store_row<TCS_DP_AUX>(rsF, vo, Q.A);
                    }
                    if (edge_lane) tcs_store_edge_fwd(rsE, ve, Q.A.c[R - 1], Q.umA, Q.e, Q.eh);
                    const Masks<R> m1 = row_masks(__builtin_amdgcn_readlane(wv16, 2 * i + 1));
                    tcs_fwd_core<SW, FLAT>(E, Q.B, Q.A, Q.carry, Q.umB, Q.umA, Q.c, m1, er + TCS_EDGE, bx, by, __builtin_amdgcn_readlane(fy.cur, yi + 1));
                    if constexpr (EM) {
                        if (lanes_of(m1.lanes)) {
                            tcs_store_row<TCS_EM_ROW_AUX>(rsF, vo + TCS_ROW_BYTES, Q.B);
                            tcs_store_planes(rsX, 2 * voff + xo + TCS_XROW_BYTES, 1024 + voff + xo + TCS_XROW_BYTES, Q.B);
                        }
                    } else {
                        if (!TCS_DP_MASK || lanes_of(m1.lanes)) tcs_store_row<TCS_DP_AUX>(rsF, vo + TCS_ROW_BYTES, Q.B);
                    }
                    if (edge_lane) tcs_store_edge_fwd(rsE, ve + 4 * TCS_EDGE, Q.B.c[R - 1], Q.umB, Q.e, Q.eh);
                    yi += 2, vo += 2 * TCS_ROW_BYTES, xo += 2 * TCS_XROW_BYTES, ve += 2 * 4 * TCS_EDGE, er += 2 * TCS_EDGE;
                }
                const Masks<R> m14 = row_masks(__builtin_amdgcn_readlane(wv16, 14));
                tcs_fwd_core<SW, FLAT>(E, Q.A, Q.B, Q.carry, Q.umA, Q.umB, Q.c, m14, er, bx, by, __builtin_amdgcn_readlane(fy.cur, yi));
                if constexpr (EM) {
                    if (lanes_of(m14.lanes)) {
                        tcs_store_row<TCS_EM_ROW_AUX>(rsF, vo, Q.A);
                        tcs_store_planes(rsX, 2 * voff + xo, 1024 + voff + xo, Q.A);
                    }
                } else {
                    if (!TCS_DP_MASK || lanes_of(m14.lanes)) tcs_store_row<TCS_DP_AUX>(rsF, vo, Q.A);
                }
                if (edge_lane) tcs_store_edge_fwd(rsE, ve, Q.A.c[R - 1], Q.umA, Q.e, Q.eh);
                wait_vm();
                if (lane == 0) lds_poke(prog + wv, static_cast<int>(st.row0) + (d + 14 - st.df) + 1);
                rm += TCS_BLOCK - 1;
                d += TCS_BLOCK - 1;
                }
                gstep(d);
                ++d;
            }
            if (s == S - 1) {  // total probability at the end corner (lX, lY), anti-diagonal D = this stripe's last row
                const int je = lX - st.X;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (jr[r] == je) {
                        const RCell c = (st.dl & 1) ? Q.A.c[r] : Q.B.c[r];
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
        const float tot_m = unif(reinterpret_cast<float *>(lmisc)[0]);
        const int tot_e = uni(lmisc[1]);

        TaskOut out;
        out.tot_m = tot_m, out.tot_e = tot_e, out.btot_m = 0.f, out.btot_e = E_DEAD, out.npairs = 0;
        out.status = NPR_OK;
        const bool alive = tot_m > 0.f;
        if (!alive) out.status = NPR_ERR_ZERO_PROB;

        float em_acc[15];  // E-step: the expected transition counts of this lane's cells (k_em_tile's order)
        int boost = 0;     // ... and how far its backward values outgrew 2^TCS_TOP inside a block
#pragma unroll
        for (int i = 0; i < 15; ++i) em_acc[i] = 0.f;
        // =============================== backward + posteriors ===============================
#ifdef NPR_TCS_EM_EXP
        if (alive && !(EM && (NPR_TCS_EM_EXP == 1 || NPR_TCS_EM_EXP == 3))) {
#else
        if (alive) {
#endif
            const float inv_tot = 1.0f / tot_m;
            const float thr_lo = a.threshold * tot_m * (1.0f - 1.0f / 1024.0f);
            const PairSink sink{a.px, a.py, a.pp, pair_off, pair_cap, xs, ys, a.threshold};
            if (threadIdx.x < NWMAX) prog[threadIdx.x] = 0x7fffffff;  // now: the LOWEST row whose neighbour cell is out
            __syncthreads();
            int vsmax = -(1 << 30);
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
                const int X0 = st.X;
                Bases<R> bx, by;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    bx.b[r] = base8<RS_XS>(E.X, lX, X0 + jr[r]);
                    by.b[r] = base8(E.Y, lY, (st.dl + 1) - X0 - jr[r]);  // as of anti-diagonal dl + 1
                }
                Feed fy;
                feed8_init<-1>(fy, E.Y, lY, st.dl - X0 - (K - 1), lane);
                RFRow<R> fa, fb;  // the forward row of an odd anti-diagonal in fa, of an even one in fb: loaded one step ahead of its use
#pragma unroll
                for (int r = 0; r < R; ++r) fa.v[r] = fb.v[r] = 0.f;
                cptr32 rm = rowmask + st.row0 + static_cast<uint32_t>(st.dl - st.df);  // the word of the row in hand is rm[0]
                const __amdgpu_buffer_rsrc_t rsF = stripe_rsrc(F, st.row0, TCS_ROW_BYTES), rsE = stripe_rsrc(Eb, st.row0, 4 * TCS_EDGE);
                const __amdgpu_buffer_rsrc_t rsX = EM ? stripe_rsrc(Fx, st.row0, TCS_XROW_BYTES) : rsF;
                if constexpr (!EM) {
                    const v2i q = __builtin_amdgcn_raw_buffer_load_b64(rsF, voff + (st.dl - st.df) * TCS_ROW_BYTES, 0, TCS_DP_AUX);
                    RFRow<R> &f0 = (st.dl & 1) ? fa : fb;
                    f0.v[0] = bitsf(q.x), f0.v[1] = bitsf(q.y);
                }
                // ---- E-step: the left stripe (its forward records feed lane 0's counts), the symbols emitted INTO every slot's cell, the bins' rows ----
                int dfL = 1, dlL = 0;
                uint32_t row0L = 0;
                Bases<R> bxm, bym;
                Feed fym;
                int binx[R];     // byte offset of the match bins' row of X[x-1] + this lane's column; N: the scratch zone
                float xbs[R], xbl[R];  // this stripe's counts of X[x-1] emitted by shortGapX / longGapX into the slot's column: one base per slot, so registers
#pragma unroll
                for (int r = 0; r < R; ++r) xbs[r] = xbl[r] = 0.f;
                if constexpr (EM) {
                    if (s > 0) {
                        const UStripe sl = load_stripe(tab, s - 1);
                        dfL = sl.df, dlL = sl.dl, row0L = sl.row0;
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int xi = X0 + jr[r] - 1;
                        bxm.b[r] = base8<RS_XS>(E.X, lX, xi);
                        bym.b[r] = base8(E.Y, lY, (st.dl + 1) - X0 - jr[r] - 1);
                        binx[r] = (bxm.b[r] >= RS_NX ? EM_BINS * 256 : base8<1024>(E.X, lX, xi)) + 4 * lane;  // (N, or no base: the scratch zone)
                    }
                    feed8_init<-1>(fym, E.Y, lY, st.dl - X0 - (K - 1) - 1, lane);
                }
                // the right stripe's block of rows [16 kb, 16 kb + 15]
                auto take_block = [&](int kb, int &e_in, int &eh_in) {
                    const int first = kb * TCS_BLOCK, lo = max(first, dfR), hi = min(first + TCS_BLOCK - 1, dlR);
                    if (hi >= lo) {  // uniform
                        const int need = static_cast<int>(row0R) + (lo - dfR);
                        tcs_wait_at_most(prog + wR, need, stuck);
                    }
                    const EdgeExp x = tcs_edge_stage(Eb, row0R, dfR, first, lo, hi, stage, lane);
                    e_in = hi >= lo ? __builtin_amdgcn_readlane(x.e, lo - first) : TCS_NONE;
                    eh_in = hi >= lo ? __builtin_amdgcn_readlane(x.eh, lo - first) : TCS_NONE;
                };
                // the forward sweep's exponents of the block that holds row d, and what turns F * B of a lane into a posterior there
                // (in two factors: a lane a value has just entered holds mantissas far above 2^TCS_TOP until its next renormalisation -- up to
                // 2^(TCS_TOP + 6 + 8 TCS_C) in each sweep --, so neither F * B nor 2^(eF + eB - eTot) alone is safe in fp32.  Both factors are
                // powers of two: F * B is still rounded once, as npr_rs.h's rs_posterior rounds it.)
                float G1 = 0.f, G2 = 0.f;
                CsState Q;
                // E-step: the forward exponents of the block that holds row d (kbCur) and of the one below it: the rows d - 1 .. d - 3 lie in one of the two
                int kbCur = 0, efCur = TCS_NONE, efPrev = TCS_NONE;
                bool cnt_on = true;
                float cb = 1.f;  // 2^(e - e of the lane below): what the lane below's scaled forward cells are multiplied by on their way up
                auto enter_block = [&](int d) __attribute__((always_inline)) {
                    const int kf = max(d & ~(TCS_BLOCK - 1), st.df) - st.df;
                    const v2i x = __builtin_amdgcn_raw_buffer_load_b64(rsF, kf * TCS_ROW_BYTES + TCS_EXP_AT + 8 * lane, 0, 0);
                    if constexpr (EM) {
                        kbCur = d >> 4;
                        efCur = x.x, efPrev = TCS_NONE;
                        int ehF = x.y;  // the bound on the forward values the rows d - 1, d - 2 of this block's rows can hold: the two blocks', and the lane below's
                        const int below = kbCur * TCS_BLOCK - 1;  // the highest row of the block below
                        if (below >= st.df) {
                            const v2i y = __builtin_amdgcn_raw_buffer_load_b64(rsF, (max(below & ~(TCS_BLOCK - 1), st.df) - st.df) * TCS_ROW_BYTES + TCS_EXP_AT + 8 * lane, 0, 0);
                            efPrev = y.x, ehF = max(ehF, y.y);
                        }
                        ehF = max(ehF, dpp_from_below(ehF, ehF));
                        cb = tcs_pow2(Q.e - dpp_from_below(Q.e, Q.e));
                        // A block in which no lane's F B / total can reach 2^TCS_EM_DEAD (values are below 2^(TCS_TOP + 6) in units of their bound) adds nothing a
                        // count can see: its rows are not counted -- where the band crosses a stripe ahead of or behind the alignment, two rows in five.
                        // (+ TCS_C: lane 0 takes the left stripe's last column, whose exponents the chain keeps within TCS_C of its own)
                        cnt_on = __ballot(ehF + TCS_C + Q.eh - tot_e + 2 * (TCS_TOP + 6) + 1 > TCS_EM_DEAD) != 0;
                    } else {
                        const int sx = x.x + Q.e - tot_e;
                        G1 = tcs_pow2(sx >> 1), G2 = tcs_pow2(sx - (sx >> 1));
                    }
                    vsmax = max(vsmax, x.y + Q.eh - tot_e);
                };
                // E-step: the forward rows d - 1 and d - 2 in this lane's posterior units -- F 2^(eF + eB - eTot) --, the lane below's top cell of
                // each (GAb, GBb; lane 0: the left stripe's last column) in the same units, and the row d - 3 in flight as loaded
                // (odd d: row d - 1 in GA, row d - 2 in GB; even d: the other way round -- the row that arrives takes the registers of the one that leaves)
                RDiag<R> GA = zero_rdiag<R>(), GB = zero_rdiag<R>();
                RCell GAb = zero_rcell(), GBb = zero_rcell();
                struct EmRow {
                    v2i f;
                    v4i x0;
                    v2i x1;
                };
                EmRow S0{v2i{0, 0}, v4i{0, 0, 0, 0}, v2i{0, 0}}, S1 = S0;  // two rows in flight: an even row lands in S0, an odd one in S1
                int kbF = -(1 << 30);  // the block of the left stripe's forward records in stageF
                // issue the loads of a row (zeros outside the stripe's rows and in the lanes that hold no band cell: those were not stored); w: the row's mask word
                auto em_fetch_w = [&](int row, uint32_t w, EmRow &S) __attribute__((always_inline)) {
                    S.f = v2i{0, 0}, S.x0 = v4i{0, 0, 0, 0}, S.x1 = v2i{0, 0};
                    if (row >= st.df && row <= st.dl) {  // uniform
                        const int k = row - st.df;
#ifdef NPR_TCS_EM_EXP
                        if (NPR_TCS_EM_EXP != 2)
#endif
                        if (lanes_of(row_masks(w).lanes)) {
                            S.f = __builtin_amdgcn_raw_buffer_load_b64(rsF, voff + k * TCS_ROW_BYTES, 0, TCS_EM_AUX);
                            S.x0 = __builtin_amdgcn_raw_buffer_load_b128(rsX, 2 * voff + k * TCS_XROW_BYTES, 0, TCS_EM_AUX);
                            S.x1 = __builtin_amdgcn_raw_buffer_load_b64(rsX, 1024 + voff + k * TCS_XROW_BYTES, 0, TCS_EM_AUX);
                        }
                    }
                };
                auto em_fetch = [&](int row, EmRow &S) __attribute__((always_inline)) {  // ... the word by a scalar load (the general step, the stripe's first rows)
                    const int k = min(max(row - st.df, 0), st.dl - st.df);
                    em_fetch_w(row, rowmask[st.row0 + static_cast<uint32_t>(k)], S);
                };
                auto em_scaled = [&](int row, const EmRow &S) __attribute__((always_inline)) -> RDiag<R> {  // a fetched row in this lane's posterior units
                    const int sF = min(max(((row >> 4) == kbCur ? efCur : efPrev) + Q.e - tot_e, -300), 300);
                    RDiag<R> G;
                    G.c[0].m = bitsf(S.f.x), G.c[1].m = bitsf(S.f.y);
                    tcs_unpack24(S.x0.x, S.x0.y, S.x0.z, G.c[0].sx, G.c[0].sy, G.c[0].lx, G.c[0].ly);
                    tcs_unpack24(S.x0.w, S.x1.x, S.x1.y, G.c[1].sx, G.c[1].sy, G.c[1].lx, G.c[1].ly);
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        G.c[r] = RCell{__builtin_ldexpf(G.c[r].m, sF), __builtin_ldexpf(G.c[r].sx, sF), __builtin_ldexpf(G.c[r].sy, sF), __builtin_ldexpf(G.c[r].lx, sF), __builtin_ldexpf(G.c[r].ly, sF)};
                    return G;
                };
                auto em_left = [&](int row) __attribute__((always_inline)) -> RCell {  // the left stripe's last column on `row`, in the posterior units of the lane that asks (lane 0 uses it)
                    if ((row >> 4) != kbF) {  // uniform: once per sixteen rows
                        kbF = row >> 4;
                        const int first = kbF * TCS_BLOCK;
                        (void)tcs_edge_stage(Ef, row0L, dfL, first, max(first, dfL), min(first + TCS_BLOCK - 1, dlL), stageF, lane);
                    }
                    const float *rec = stageF + TCS_EDGE * (row & (TCS_BLOCK - 1));
                    const float4 q = *reinterpret_cast<const float4 *>(rec), g = *reinterpret_cast<const float4 *>(rec + 4);
                    const int sF = min(max(fbits(g.y) + Q.e - tot_e, -300), 300);
                    return RCell{__builtin_ldexpf(q.y, sF), __builtin_ldexpf(q.z, sF), __builtin_ldexpf(g.x, sF), __builtin_ldexpf(q.w, sF), __builtin_ldexpf(g.w, sF)};
                };
                auto em_take = [&](int row, const EmRow &S, RDiag<R> &G, RCell &Gbelow) __attribute__((always_inline)) {  // a fetched row, and the lane below's top cell of it in this lane's units (lane 0: cb = 1)
                    G = em_scaled(row, S);
                    Gbelow = dpp_rcell_from_below(G.c[R - 1], em_left(row));
                    rcell_scale(Gbelow, cb);
                };
                {
                    int e_in, eh_in;
                    take_block((st.dl + 1) >> 4, e_in, eh_in);
                    tcs_init<false>(Q, lane, e_in, eh_in, s == S - 1 ? (lX - X0) / R : WAVE, dslot);  // (the end cell (lX, lY) lies in the last stripe)
                    // (x+1, y+1) of the top slot on the first anti-diagonal: the right stripe's cell on dl + 2
                    const int q0 = st.dl + 2;
                    if (q0 >= dfR && q0 <= dlR) {  // uniform
                        const int need = static_cast<int>(row0R) + (q0 - dfR);
                        tcs_wait_at_most(prog + wR, need, stuck);
                        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(Eb + (static_cast<int64_t>(row0R) + (q0 - dfR)) * (4 * TCS_EDGE), 0, -1, 0x00020000);
                        const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rq, 0, 0, 16);
                        const v4i g = __builtin_amdgcn_raw_buffer_load_b128(rq, 16, 0, 16);
                        if (lane == WAVE - 1) Q.carry = __builtin_ldexpf(bitsf(q.x), g.y - Q.e);  // (the record's m, in this lane's units)
                    }
                    enter_block(st.dl);
                    if constexpr (EM) {
                        const int ra = (st.dl & 1) ? st.dl - 1 : st.dl - 2;  // (GA holds the row of an even anti-diagonal, GB of an odd one)
                        em_fetch(ra, S0);
                        em_take(ra, S0, GA, GAb);
                        em_fetch(2 * st.dl - 3 - ra, S1);  // (the odd one of dl - 1, dl - 2)
                        em_take(2 * st.dl - 3 - ra, S1, GB, GBb);
                        if (st.dl & 1) em_fetch(st.dl - 3, S0);  // the row the first step takes: fetched two steps ahead from here on
                        else em_fetch(st.dl - 3, S1);
                    }
                }
                // E-step: the expected counts of the transitions into the cells of anti-diagonal d (k_em_tile's tile_em_cells, in this arithmetic)
                // P1 / P1b: row d - 1 and its lane-below cell, P2 / P2b: row d - 2's.  An accumulator adds up F' * w of its transition (the transition itself
                // multiplies the sum once, at the end of the task); a bin gets w * (the recurrence's sum over the states the symbol can be emitted from).
                auto em_count = [&](const RDiag<R> &io, const RDiag<R> &P1, const RCell &P1b, const RDiag<R> &P2, const RCell &P2b) __attribute__((always_inline)) {
                    float em[R], exs[R], exl[R], eys[R], eyl[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) rs_cell_emissions<4, FLAT>(E.ltab, ~0ull, bxm.b[r], bym.b[r], em[r], exs[r], exl[r], eys[r], eyl[r]);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const RCell &c = io.c[r];
                        const RCell &Pm = r ? P2.c[r - 1] : P2b, &Pl = r ? P1.c[r - 1] : P1b, &Pu = P1.c[r];
                        const Trans &t = E.tr;
                        float bM, bYs, bYl;
                        {
                            const float w = em[r] * c.m * inv_tot;
                            em_acc[0] = __builtin_fmaf(Pm.m, w, em_acc[0]), em_acc[1] = __builtin_fmaf(Pm.sx, w, em_acc[1]), em_acc[2] = __builtin_fmaf(Pm.sy, w, em_acc[2]);
                            em_acc[3] = __builtin_fmaf(Pm.lx, w, em_acc[3]), em_acc[4] = __builtin_fmaf(Pm.ly, w, em_acc[4]);
                            bM = tcs_into_match(t, Pm) * w;
                        }
                        {
                            const float ws = exs[r] * c.sx * inv_tot, wl = exl[r] * c.lx * inv_tot;
                            em_acc[5] = __builtin_fmaf(Pl.m, ws, em_acc[5]), em_acc[6] = __builtin_fmaf(Pl.sx, ws, em_acc[6]);
                            em_acc[8] = __builtin_fmaf(Pl.m, wl, em_acc[8]), em_acc[9] = __builtin_fmaf(Pl.lx, wl, em_acc[9]);
                            float as = __builtin_fmaf(t.sxsx, Pl.sx, t.msx * Pl.m);
                            if constexpr (SW) em_acc[7] = __builtin_fmaf(Pl.sy, ws, em_acc[7]), as = __builtin_fmaf(t.sysx, Pl.sy, as);
                            xbs[r] = __builtin_fmaf(as, ws, xbs[r]);
                            xbl[r] = __builtin_fmaf(__builtin_fmaf(t.lxlx, Pl.lx, t.mlx * Pl.m), wl, xbl[r]);
                        }
                        {
                            const float ws = eys[r] * c.sy * inv_tot, wl = eyl[r] * c.ly * inv_tot;
                            em_acc[10] = __builtin_fmaf(Pu.m, ws, em_acc[10]), em_acc[11] = __builtin_fmaf(Pu.sy, ws, em_acc[11]);
                            em_acc[13] = __builtin_fmaf(Pu.m, wl, em_acc[13]), em_acc[14] = __builtin_fmaf(Pu.ly, wl, em_acc[14]);
                            float as = __builtin_fmaf(t.sysy, Pu.sy, t.msy * Pu.m);
                            if constexpr (SW) em_acc[12] = __builtin_fmaf(Pu.sx, ws, em_acc[12]), as = __builtin_fmaf(t.sxsy, Pu.sx, as);
                            bYs = as * ws;
                            bYl = __builtin_fmaf(t.lyly, Pu.ly, t.mly * Pu.m) * wl;
                        }
                        // the bins the read's base decides (k_em_tile's layout; the long-gap bin 4 rows after the short-gap one): reads, then writes
                        constexpr int TRASH = EM_BINS * 256;
                        const int ey4 = bym.b[r];  // (the read's codes come scaled by 4: RS_YS; 16: N)
                        const bool ny = ey4 >= 16;
                        const int aM = (ny ? TRASH : ey4 * 64) + binx[r];  // (binx of an N: beyond every bin row, so the sum is clamped into the scratch zone)
                        const int aMc = min(aM, TRASH + 4 * lane);
                        const int aY = (ny ? TRASH : 24 * 256 + ey4 * 64) + 4 * lane;
                        char *const lb = reinterpret_cast<char *>(lbins);
                        const float v0 = *reinterpret_cast<float *>(lb + aMc), v3 = *reinterpret_cast<float *>(lb + aY), v4 = *reinterpret_cast<float *>(lb + aY + 1024);
                        *reinterpret_cast<float *>(lb + aMc) = v0 + bM;
                        *reinterpret_cast<float *>(lb + aY) = v3 + bYs;
                        *reinterpret_cast<float *>(lb + aY + 1024) = v4 + bYl;
                    }
                };
                // posteriors of anti-diagonal d, slots claimed from the workgroup's LDS counter
                auto emit = [&](int d, const RDiag<R> &io, const RFRow<R> &f, const Masks<R> &mk) {
                    float q[R];
                    uint64_t cand[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        q[r] = (f.v[r] * G1) * (io.c[r].m * G2);
                        cand[r] = __ballot(q[r] >= thr_lo) & mk.cell[r];
                    }
                    if (d >= 2 && (cand[0] | cand[1])) {
                        float p[R];
                        uint64_t hit[R];
                        int total = 0;
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            p[r] = q[r] * inv_tot;
                            hit[r] = __ballot(p[r] >= sink.threshold) & cand[r];
                            total += __popcll(hit[r]);
                        }
                        if (total) {
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
                };

                // any row.  f: the forward row of d (loaded a step ago); fnext: where the row of d-1 goes
                auto step = [&](int d, RDiag<R> &io, const RDiag<R> &s1, RFRow<R> &f, RFRow<R> &fnext, RDiag<R> &P1, RCell &P1b, const RDiag<R> &P2, const RCell &P2b) __attribute__((always_inline)) {
                    const int k = d - st.df;
                    const Masks<R> mk = row_masks(rm[0]);
                    if (d > st.df) {
                        rm -= 1;
                        if constexpr (!EM) {
                            const v2i q = __builtin_amdgcn_raw_buffer_load_b64(rsF, voff + (k - 1) * TCS_ROW_BYTES, 0, TCS_DP_AUX);
                            fnext.v[0] = bitsf(q.x), fnext.v[1] = bitsf(q.y);
                        }
                    }
                    if constexpr (EM) {
                        em_fetch(d - 4, S1);  // (d is odd here: taken at the end of the NEXT step -- two rows in flight)
                        bases_up<R>(bym, feed8_get<-1>(fym, E.Y, lY, d - X0 - (K - 1) - 1, lane));
                    }
                    tcs_bwd_core<SW, FLAT>(E, io, s1, Q.carry, Q.c, mk, stage + TCS_EDGE * ((d + 1) & (TCS_BLOCK - 1)), bx, by, feed8_get<-1>(fy, E.Y, lY, d - X0 - (K - 1), lane));
                    if (d == D) {  // the end corner (lX, lY)
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            if (X0 + jr[r] == lX) {
                                const int k0 = -Q.e;
                                io.c[r] = RCell{__builtin_ldexpf(mdl->end[re * 5 + 0], k0), __builtin_ldexpf(mdl->end[re * 5 + 1], k0), __builtin_ldexpf(mdl->end[re * 5 + 2], k0),
                                                __builtin_ldexpf(mdl->end[re * 5 + 3], k0), __builtin_ldexpf(mdl->end[re * 5 + 4], k0)};
                            }
                    }
                    if ((d & (TCS_BLOCK - 1)) == TCS_BLOCK - 1) {  // the rows from d down share new exponents; the right stripe's block of the steps to come
                        if (d > st.df) {
                            int e_in, eh_in;
                            take_block(d >> 4, e_in, eh_in);
                            const int e_was = Q.e;
                            const int eb = tcs_renorm<false>(Q, lane, e_in, eh_in, dslot);
                            if constexpr (EM) {  // the held forward rows are in units of 2^(eF + eB - eTot): eB has moved
                                boost = max(boost, eb - (126 + TCS_TOP));
                                const float gf = tcs_pow2(Q.e - e_was);
#pragma unroll
                                for (int r = 0; r < R; ++r) rcell_scale(GA.c[r], gf), rcell_scale(GB.c[r], gf);
                                rcell_scale(GAb, gf), rcell_scale(GBb, gf);
                            }
                        }
                        enter_block(d);
                    }
                    if (lane == 0) tcs_store_edge_bwd(rsE, 4 * TCS_EDGE * k, io.c[0], Q.e, Q.eh);  // holds the stripe's first column in its register 0
                    if constexpr (EM) {
                        if (cnt_on) em_count(io, P1, P1b, P2, P2b);
                        em_take(d - 3, S0, P1, P1b);  // (into the registers of the row that leaves)
                    } else {
                        emit(d, io, f, mk);
                    }
                    if ((d & (TCS_BLOCK - 1)) == 0 || d == st.df) {
                        wait_vm();
                        if (lane == 0) lds_poke(prog + wv, static_cast<int>(st.row0) + k);
                    }
                };
                auto gstep = [&](int d) {
                    if (!(d & 1)) {
                        tcs_swap_rows(Q.A, Q.B);
                        const RFRow<R> t = fa;
                        fa = fb, fb = t;
                        if constexpr (EM) {  // (the held rows and the rows in flight follow the parity too)
                            tcs_swap_rows(GA, GB);
                            const RCell u = GAb;
                            GAb = GBb, GBb = u;
                            const EmRow v = S0;
                            S0 = S1, S1 = v;
                        }
                    }
                    step(d, Q.A, Q.B, fa, fb, GA, GAb, GB, GBb);
                    if (!(d & 1)) {
                        if constexpr (EM) {
                            tcs_swap_rows(GA, GB);
                            const RCell u = GAb;
                            GAb = GBb, GBb = u;
                            const EmRow v = S0;
                            S0 = S1, S1 = v;
                        }
                        tcs_swap_rows(Q.A, Q.B);
                        const RFRow<R> t = fa;
                        fa = fb, fb = t;
                    }
                };
                // one loop, one instance of the general step; where a row 16 kb + 14 opens a whole block inside the stripe (the end cell's row never does), the fast loop
                // takes the fifteen rows 16 kb + 14 .. 16 kb (d is even: pairs of a B step and an A step) and the general step the boundary row 16 kb - 1
                int d = st.dl;
                while (d >= st.df) {
                    if ((d & (TCS_BLOCK - 1)) == TCS_BLOCK - 2 && d - (TCS_BLOCK - 1) >= st.df && d < D) {
                    feed8_ahead<-1>(fy, E.Y, lY, d - X0 - (K - 1), lane);
                    int yi = uni(fy.base - (d - X0 - (K - 1)));
                    int vo = voff + (d - 1 - st.df) * TCS_ROW_BYTES, ve = 4 * TCS_EDGE * (d - st.df);  // vo: the row loaded ahead, d - 1
                    const float *er = stage + TCS_EDGE * (TCS_BLOCK - 1);  // record (d + 1) & 15 = 15
                    // the mask words of rows d .. d - 15: lane i (mod 16) holds row d - i's
                    // (E-step: of rows d .. d - 31, lane i mod 32 -- the rows d - 3 .. d - 17 it loads are masked by theirs; below the stripe's first row: unused)
                    const int wv16 = static_cast<int>(a.rowmask[rowmask_off + st.row0 + static_cast<uint32_t>(max(d - st.df - (lane & (EM ? 2 * TCS_BLOCK - 1 : TCS_BLOCK - 1)), 0))]);
#pragma unroll 1
                    for (int i = 0; i < 7; ++i) {
                        if constexpr (EM) {  // (the symbols emitted into the cells: the read's bases one step further along the window)
                            em_fetch_w(d - 2 * i - 4, __builtin_amdgcn_readlane(wv16, 2 * i + 4), S0);
                            bases_up<R>(bym, __builtin_amdgcn_readlane(fy.cur, yi + 1));
                        } else {
                            if (!TCS_DP_MASK || lanes_of(row_masks(__builtin_amdgcn_readlane(wv16, 2 * i + 1)).lanes)) {
                                const v2i q = __builtin_amdgcn_raw_buffer_load_b64(rsF, vo, 0, TCS_DP_AUX);
                                fa.v[0] = bitsf(q.x), fa.v[1] = bitsf(q.y);
                            }
                        }
                        const Masks<R> m0 = row_masks(__builtin_amdgcn_readlane(wv16, 2 * i));
                        tcs_bwd_core<SW, FLAT>(E, Q.B, Q.A, Q.carry, Q.c, m0, er, bx, by, __builtin_amdgcn_readlane(fy.cur, yi));
                        if (lane == 0) tcs_store_edge_bwd(rsE, ve, Q.B.c[0], Q.e, Q.eh);
                        if constexpr (EM) {
                            if (cnt_on) em_count(Q.B, GB, GBb, GA, GAb);
                            em_take(d - 2 * i - 3, S1, GB, GBb);
                            em_fetch_w(d - 2 * i - 5, __builtin_amdgcn_readlane(wv16, 2 * i + 5), S1);
                            bases_up<R>(bym, __builtin_amdgcn_readlane(fy.cur, yi + 2));
                        } else {
                            emit(d - 2 * i, Q.B, fb, m0);
                            if (!TCS_DP_MASK || lanes_of(row_masks(__builtin_amdgcn_readlane(wv16, 2 * i + 2)).lanes)) {
                                const v2i q = __builtin_amdgcn_raw_buffer_load_b64(rsF, vo - TCS_ROW_BYTES, 0, TCS_DP_AUX);
                                fb.v[0] = bitsf(q.x), fb.v[1] = bitsf(q.y);
                            }
                        }
                        const Masks<R> m1 = row_masks(__builtin_amdgcn_readlane(wv16, 2 * i + 1));
                        tcs_bwd_core<SW, FLAT>(E, Q.A, Q.B, Q.carry, Q.c, m1, er - TCS_EDGE, bx, by, __builtin_amdgcn_readlane(fy.cur, yi + 1));
                        if (lane == 0) tcs_store_edge_bwd(rsE, ve - 4 * TCS_EDGE, Q.A.c[0], Q.e, Q.eh);
                        if constexpr (EM) {
                            if (cnt_on) em_count(Q.A, GA, GAb, GB, GBb);
                            em_take(d - 2 * i - 4, S0, GA, GAb);
                        } else {
                            emit(d - 2 * i - 1, Q.A, fa, m1);
                        }
                        yi += 2, vo -= 2 * TCS_ROW_BYTES, ve -= 2 * 4 * TCS_EDGE, er -= 2 * TCS_EDGE;
                    }
                    {
                        if constexpr (EM) {
                            em_fetch_w(d - 18, __builtin_amdgcn_readlane(wv16, 18), S0);
                            bases_up<R>(bym, __builtin_amdgcn_readlane(fy.cur, yi + 1));
                        } else {
                            if (!TCS_DP_MASK || lanes_of(row_masks(__builtin_amdgcn_readlane(wv16, 15)).lanes)) {
                                const v2i q = __builtin_amdgcn_raw_buffer_load_b64(rsF, vo, 0, TCS_DP_AUX);
                                fa.v[0] = bitsf(q.x), fa.v[1] = bitsf(q.y);
                            }
                        }
                        const Masks<R> m0 = row_masks(__builtin_amdgcn_readlane(wv16, 14));
                        tcs_bwd_core<SW, FLAT>(E, Q.B, Q.A, Q.carry, Q.c, m0, er, bx, by, __builtin_amdgcn_readlane(fy.cur, yi));
                        if (lane == 0) tcs_store_edge_bwd(rsE, ve, Q.B.c[0], Q.e, Q.eh);
                        if constexpr (EM) {
                            if (cnt_on) em_count(Q.B, GB, GBb, GA, GAb);
                            em_take(d - 17, S1, GB, GBb);
                        } else {
                            emit(d - 14, Q.B, fb, m0);
                        }
                        wait_vm();
                        if (lane == 0) lds_poke(prog + wv, static_cast<int>(st.row0) + (d - 14 - st.df));
                    }
                    rm -= TCS_BLOCK - 1;
                    d -= TCS_BLOCK - 1;
                    if constexpr (EM) feed8_init<-1>(fym, E.Y, lY, d - X0 - (K - 1) - 1, lane);  // (the general step's own window of those bases, where the block left off)
                    }
                    gstep(d);
                    --d;
                }
                if constexpr (EM) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {  // this stripe's gap-X emission counts into the bins of its slots' bases (binx: the match row of the base, 4 bins wide)
                        float *const at = reinterpret_cast<float *>(reinterpret_cast<char *>(lbins) + (binx[r] - 4 * lane >= EM_BINS * 256 ? EM_BINS * 256 : 16 * 256 + ((binx[r] - 4 * lane) >> 2)) + 4 * lane);
                        at[0] += xbs[r];
                        at[256] += xbl[r];
                    }
                }
                if (s == 0) {  // total from the backward side: the lattice point (0, 0) is the stripe's first slot on d = 0
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (X0 + jr[r] == 0) {
                            const RCell cz = (st.df & 1) ? Q.A.c[r] : Q.B.c[r];
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
            atomicMax(lmisc + 6, vsmax);
            if constexpr (EM)
                if (boost > TCS_EM_BOOST) atomicMax(lmisc + 6, (1 << 29) + boost);
            __syncthreads();
            out.btot_m = unif(reinterpret_cast<float *>(lmisc)[2]);
            out.btot_e = uni(lmisc[3]);
        }
        if (stuck) atomicMax(lmisc + 6, 1 << 30);
        __syncthreads();
        if constexpr (EM) {
            // The counts of a task leave the workgroup only when its certificate holds and every sum is a finite number (else k_em_tile counts the task:
            // TASK_RERUN): every wavefront adds up its own bins, then its transition accumulators through the same rows (k_em_tile's reduction).
            const bool refused = a.wcap == 1 && (t & 1);  // NPR_OPT_EM_TILE = 2 (tests): every other task goes to the second pass whatever its certificate says
            const bool cert = alive && uni(lmisc[6]) < TCS_S_LIMIT && !refused;
            double sumE = 0.0, sumT = 0.0;
            const int map[15] = {0, 5, 10, 15, 20, 1, 6, 11, 3, 18, 2, 12, 7, 4, 24};  // accumulator order -> T[from*5+to]
            if (cert) {
                if (lane < EM_BINS)
                    for (int q = 0; q < WAVE; ++q) sumE += static_cast<double>(lbins[lane * WAVE + q]);
#pragma unroll
                for (int i = 0; i < 15; ++i) lbins[i * WAVE + lane] = em_acc[i];  // (sums of F' * w: the transition multiplies below)
                if (lane < 15) {
                    for (int q = 0; q < WAVE; ++q) sumT += static_cast<double>(lbins[lane * WAVE + q]);
                    sumT *= static_cast<double>(mdl->T[map[lane]]);
                }
                const double big = 1e300;
                if (__ballot(!(sumE > -big && sumE < big && sumT > -big && sumT < big)) != 0 && lane == 0) atomicMax(lmisc + 6, 1 << 30);
            }
            if (refused && threadIdx.x == 0) atomicMax(lmisc + 6, 1 << 30);
            __syncthreads();
            if (cert && uni(lmisc[6]) < TCS_S_LIMIT) {
                if (lane < EM_BINS) atomicAdd(a.em_E + model * EM_BINS + lane, sumE);
                if (lane < 15) atomicAdd(a.em_T + model * 25 + map[lane], sumT);
            }
        }
        if (threadIdx.x == 0) {
            const int cnt = lmisc[4];
            out.npairs = cnt;
            if (cnt > pair_cap) out.status = NPR_ERR_CAPACITY;
            if (!alive || lmisc[6] >= TCS_S_LIMIT) out.status = TASK_RERUN, out.npairs = lmisc[6];  // one exponent per lane may not have been enough (or nothing arrived: k_dp_tile decides)
            a.outs[t] = out;
            lmisc[5] = atomicAdd(a.queue, 1);
        }
        __syncthreads();
        t = uni(lmisc[5]) + static_cast<int>(gridDim.x);
    }
}

}  // namespace

size_t tile_cs_lds_bytes(int) { return 0; }  // (static LDS: sizeof(CsLds))

int launch_tile_cs(const KernelArgs &a, int NW, int grid, void *stream, bool sw, bool flat) {
    if (NW < 1 || NW > TCS_MAX_NW) return static_cast<int>(hipErrorInvalidValue);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (sw) hipLaunchKernelGGL((k_dp_tile_cs<true, false>), dim3(grid), dim3(WAVE * NW), 0, s, a);
    else if (flat) hipLaunchKernelGGL((k_dp_tile_cs<false, true>), dim3(grid), dim3(WAVE * NW), 0, s, a);
    else hipLaunchKernelGGL((k_dp_tile_cs<false, false>), dim3(grid), dim3(WAVE * NW), 0, s, a);
    return static_cast<int>(hipGetLastError());
}


// k_dp_tile_cs<.., EM>: the E-step on the same stripes (static LDS; TCS_EM_NW wavefronts per task at most)
int em_tile_cs_waves() {  // wavefronts per task (NPR_EM_CS_WAVES: bring-up)
    const char *e = std::getenv("NPR_EM_CS_WAVES");
    const int n = e ? std::atoi(e) : TCS_EM_NW;
    return n >= 1 && n <= TCS_EM_NW ? n : TCS_EM_NW;
}
int em_tile_cs_waves_per_cu() {  // what the registers (NPR_TCS_EM_WAVES per SIMD) and the workgroups' static LDS leave room for
    const int by_lds = static_cast<int>((160 * 1024) / (sizeof(CsLds<TCS_EM_NW>) + sizeof(CsEmLds))) * em_tile_cs_waves();
    return std::min(4 * NPR_TCS_EM_WAVES, by_lds);
}
int launch_em_tile_cs(const KernelArgs &a, int NW, int grid, void *stream, bool sw, bool flat) {
    if (NW < 1 || NW > TCS_EM_NW) return static_cast<int>(hipErrorInvalidValue);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (sw) hipLaunchKernelGGL((k_dp_tile_cs<true, false, true>), dim3(grid), dim3(WAVE * NW), 0, s, a);
    else if (flat) hipLaunchKernelGGL((k_dp_tile_cs<false, true, true>), dim3(grid), dim3(WAVE * NW), 0, s, a);
    else hipLaunchKernelGGL((k_dp_tile_cs<false, false, true>), dim3(grid), dim3(WAVE * NW), 0, s, a);
    return static_cast<int>(hipGetLastError());
}

}  // namespace npr
