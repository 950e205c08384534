// npr_rs.h -- the ROW-SCALED arithmetic of the one-wavefront frame kernels (k_dp_rs<R>, npr_kernel_rs.hip; round 3).
//
// npr_cell.h carries one binary exponent per CELL next to its five mantissas.  That costs about 20 of a cell's ~55 vector
// instructions per sweep direction (three scale factors built from exponent differences, the reference-exponent maximum,
// the per-cell renormalisation, the exponent's own neighbour move and its band mask) and half of the forward row's bytes.
// A wavefront that holds a whole band row of an anti-diagonal can share ONE exponent per row instead -- the classic
// scaled forward / backward recurrence of an HMM (cactus_realign's five-state machine, SURVEY.md 8a rows a5.3-a5.5,
// reference call sites nanopore/analyses/utils.py:587, alignmentUncertainty.py:41, marginAlignSnpCaller.py:136-146):
//   * a cell is five plain fp32 values relative to 2^e, e wave-uniform (an SGPR); the two anti-diagonals held in registers
//     always share the same e, so the recurrence is 18 (forward) / 20 (backward) multiplies and FMAs and nothing else;
//   * after every RS_K-th anti-diagonal (d % RS_K == 0) the largest value of the two held rows is found (integer maximum of
//     the bit patterns -- the values are non-negative --, six DPP steps across the wavefront), both rows are multiplied by the
//     power of two that brings it to 2^85, near the top of fp32's range, and e moves by as much: a cell stays a normal
//     number down to 211 binary orders below its row's maximum (234 with denormals), which bounds the bands this
//     arithmetic is used for (rs_band_limit) -- below that a cell flushes to zero;
//   * slots outside the band hold exact zeros, kept so by computing a row under the band's lane mask (EXEC) and clearing
//     the slots a band edge has left behind -- no per-cell select;
//   * a forward row goes to the scratch as 4 bytes per cell (the match value) and the row exponents as one word per
//     RS_K rows; the backward sweep multiplies F * B * 2^(eF + eB - eTot) / totMant with the exponent sum on the scalar unit.
// oracle/realign_oracle_rs.c restates this sequence operation by operation; the parity tests demand identical bits.
#pragma once
#include <hip/hip_runtime.h>

#include "npr_frame.h"

namespace npr {

namespace {

#ifndef NPR_RS_T_SGPR_MIN_R
#define NPR_RS_T_SGPR_MIN_R 4  // transitions in SGPRs from this many slots per lane on (below: VGPRs)
#endif
constexpr int RS_K = NPR_RS_K;  // npr_device.h: shared with the host (scratch layout) and restated by the mirror
static_assert(RS_K >= 2 && (RS_K & (RS_K - 1)) == 0 && (RS_K & 1) == 0, "renormalising rows must be even anti-diagonals");

struct RCell {
    float m, sx, sy, lx, ly;
};
template <int R>
struct RDiag {
    RCell c[R];
};
__device__ __forceinline__ RCell zero_rcell() { return RCell{0.f, 0.f, 0.f, 0.f, 0.f}; }
template <int R>
__device__ __forceinline__ RDiag<R> zero_rdiag() {
    RDiag<R> d;
#pragma unroll
    for (int r = 0; r < R; ++r) d.c[r] = zero_rcell();
    return d;
}

// out[j] = in[j+1] / in[j-1]; the vacated edge slot takes 0
template <int R>
__device__ __forceinline__ RDiag<R> rs_shift_up(const RDiag<R> &in) {
    RDiag<R> o;
#pragma unroll
    for (int r = 0; r + 1 < R; ++r) o.c[r] = in.c[r + 1];
    o.c[R - 1].m = dppf_from_above(in.c[0].m);
    o.c[R - 1].sx = dppf_from_above(in.c[0].sx);
    o.c[R - 1].sy = dppf_from_above(in.c[0].sy);
    o.c[R - 1].lx = dppf_from_above(in.c[0].lx);
    o.c[R - 1].ly = dppf_from_above(in.c[0].ly);
    return o;
}
template <int R>
__device__ __forceinline__ RDiag<R> rs_shift_down(const RDiag<R> &in) {
    RDiag<R> o;
#pragma unroll
    for (int r = 1; r < R; ++r) o.c[r] = in.c[r - 1];
    o.c[0].m = dppf_from_below(in.c[R - 1].m);
    o.c[0].sx = dppf_from_below(in.c[R - 1].sx);
    o.c[0].sy = dppf_from_below(in.c[R - 1].sy);
    o.c[0].lx = dppf_from_below(in.c[R - 1].lx);
    o.c[0].ly = dppf_from_below(in.c[R - 1].ly);
    return o;
}

// ---- base streams: codes pre-multiplied by the byte strides of the tables below; code 4 is N ----
//      The read's codes (Y) by 4, the reference's (X) by 24 = 6 * 4: the match emission of (x, y) is then at byte offset bx + by of
//      em4 -- one add per cell instead of a multiply and an add.  (Round 3-4: 8 and 48, an 8-byte stride; round 5's counters showed
//      two conflict cycles per LDS instruction once the match emission was the only look-up left -- at eight bytes per entry the 25
//      live entries share 16 of the 32 banks; at four bytes each has its own.)
constexpr int RS_YS = 4, RS_XS = 24;
constexpr int RS_N8 = 4 * RS_YS, RS_NX = 4 * RS_XS;
template <int S = RS_YS>
__device__ __forceinline__ int base8(const uint8_t *seq, int len, int idx) {
    return (idx >= 0 && idx < len) ? S * static_cast<int>(seq[idx]) : 4 * S;
}
template <int DIR, int S = RS_YS>
__device__ __forceinline__ void feed8_init(Feed &f, const uint8_t *seq, int len, int first, int lane) {
    f.base = first;
    f.cur = base8<S>(seq, len, first + DIR * lane);
    f.nxt = base8<S>(seq, len, first + DIR * (64 + lane));
}
template <int DIR, int S = RS_YS>
__device__ __forceinline__ int feed8_get(Feed &f, const uint8_t *seq, int len, int idx, int lane) {
    int off = uni(DIR * (idx - f.base));
    if (off >= 64) {  // uniform
        f.cur = f.nxt;
        f.base += DIR * 64;
        f.nxt = base8<S>(seq, len, f.base + DIR * (64 + lane));
        off -= 64;
    }
    return __builtin_amdgcn_readlane(f.cur, off);
}
// ... and for the sweeps that run in blocks of RS_K anti-diagonals (NPR_RS_BLOCK): the window is looked after once per block --
// brought to where every base the block can ask for lies in `cur` -- and the steps read it without a test.  A block moves a stream's
// index by at most RS_K / 2 own steps + RS_K rebases (and the rebase statements look one base further, into `nxt` if need be).
constexpr int RS_FEED_BACK = 4;                               // bases kept behind the index (a frame that steps back; today's refill keeps none)
constexpr int RS_FEED_MAX0 = 63 - (RS_K / 2 + RS_K) + 1;      // largest offset a block may start with
template <int DIR, int S = RS_YS>
__device__ __forceinline__ void feed8_ahead(Feed &f, const uint8_t *seq, int len, int idx, int lane) {
    const int off = uni(DIR * (idx - f.base));
    if (off > RS_FEED_MAX0 || off < 0) {  // uniform
        f.base = idx - DIR * RS_FEED_BACK;
        f.cur = base8<S>(seq, len, f.base + DIR * lane);
        f.nxt = base8<S>(seq, len, f.base + DIR * (64 + lane));
    }
}
template <int DIR, int S = RS_YS, bool CHK = true>
__device__ __forceinline__ int feed8_take(Feed &f, const uint8_t *seq, int len, int idx, int lane) {
    if constexpr (CHK) return feed8_get<DIR, S>(f, seq, len, idx, lane);
    else return __builtin_amdgcn_readlane(f.cur, uni(DIR * (idx - f.base)));
}

// ---- emission tables in LDS, laid out for byte offsets that are scaled base codes ----
//   em4[6x + y] (4-byte stride: byte offset bx + by; the 25 entries of real bases and N lie in 25 different banks),
//   ex2[x] = (shortGapX, longGapX) at byte offset bx (24-byte stride), eys[y] / eyl[y] = shortGapY / longGapY at byte offset by
//   Code 5 (byte offsets RS_DEAD8 / RS_DEADX) is the base of a slot OUTSIDE the band: all its emissions are 0, so every state of the cell
//   computed there is an exact zero (two selects per cell instead of an EXEC-mask region per cell row and the clearing of what the band left behind; NPR_RS_DEADCODE_MAX_R).
struct RsTables {
    float em4[36];            // [6 x + y]
    float ex2[6][RS_XS / 4];  // [x][0 .. 1]
    float eys[6], eyl[6];
};
constexpr int RS_DEAD8 = 5 * RS_YS, RS_DEADX = 5 * RS_XS;
constexpr int RS_ZERO_EM = RS_DEADX;  // byte offset of an entry of em4 that is 0 and shares its bank with no live entry: (x = 5, y = 0), entry 30
#ifndef NPR_RS_DEADCODE_MAX_R
#define NPR_RS_DEADCODE_MAX_R 2  // slots per lane up to which it is used: one cell per lane gains 4 % (config 2: 1.75 -> 1.69 ms); two lost 1 % in round 3 and
                                 // gain since round 4 (no switch terms, seven wavefronts per SIMD): a 1/8 shard of config 3 -- a launch that is its longest read's
                                 // serial chain -- 44.5 -> 42.1 ms (two lane-mask regions and their branches fewer per step), the headline batch 139.0 -> 138.5
#endif
constexpr int RS_TABLE_FLOATS = sizeof(RsTables) / sizeof(float);
__device__ __forceinline__ void rs_build_tables(RsTables *t, const DevModel *m, int tid, int nthreads) {
    for (int i = tid; i < 36; i += nthreads) {
        const int x = i / 6, y = i % 6;
        t->em4[i] = (x < 5 && y < 5) ? m->em[5 * x + y] : 0.f;
    }
    for (int i = tid; i < 6; i += nthreads) {
        t->ex2[i][0] = i < 5 ? m->ex[5 + i] : 0.f, t->ex2[i][1] = i < 5 ? m->ex[15 + i] : 0.f;
        t->eys[i] = i < 5 ? m->ey[10 + i] : 0.f, t->eyl[i] = i < 5 ? m->ey[20 + i] : 0.f;
    }
}
__device__ __forceinline__ void rs_emissions(const char *tab, int bx, int by, float &em, float &exs, float &exl, float &eys, float &eyl) {
    constexpr int OFF_EX = offsetof(RsTables, ex2), OFF_EYS = offsetof(RsTables, eys), OFF_EYL = offsetof(RsTables, eyl);
    em = *reinterpret_cast<const float *>(tab + (bx + by));
    const float2 ex = *reinterpret_cast<const float2 *>(tab + OFF_EX + bx);
    exs = ex.x, exl = ex.y;
    eys = *reinterpret_cast<const float *>(tab + OFF_EYS + by), eyl = *reinterpret_cast<const float *>(tab + OFF_EYL + by);
}

// The emissions of one cell of a step.  Slots outside the band: up to NPR_RS_DEADCODE_MAX_R slots per lane they take the dead base code (every
// emission 0: the cell comes out as exact zeros), above that the step runs under the band's lane mask.
// FLAT: every loaded model emits every base from every gap state with probability exactly 2^-2 (all shipped ones do: blasr_hmm_0 / _20 / _40;
// a model trained by EM does not).  The four gap emissions then carry one bit -- in the band or not -- and come from a select instead of two
// 8-byte LDS loads per cell and direction; the same factor 0.25f or 0.f multiplies the same sums, so not a bit changes.
template <int R, bool FLAT>
__device__ __forceinline__ void rs_cell_emissions(const char *tab, uint64_t in_band, int bx, int by, float &em, float &exs, float &exl, float &eys, float &eyl) {
    if constexpr (R <= NPR_RS_DEADCODE_MAX_R) {
        if constexpr (FLAT) {
            const int at = bx + by;
            em = *reinterpret_cast<const float *>(tab + (lanes_of(in_band) ? at : RS_ZERO_EM));  // (every slot outside the band reads ONE zero entry: a broadcast)
            exs = exl = eys = eyl = lanes_of(in_band) ? 0.25f : 0.f;
        } else {
            rs_emissions(tab, lanes_of(in_band) ? bx : RS_DEADX, lanes_of(in_band) ? by : RS_DEAD8, em, exs, exl, eys, eyl);
        }
    } else {
        if constexpr (FLAT) {
            em = *reinterpret_cast<const float *>(tab + (bx + by));
            exs = exl = eys = eyl = 0.25f;
        } else {
            rs_emissions(tab, bx, by, em, exs, exl, eys, eyl);
        }
    }
}

// ---- the recurrence (same operand order as npr_cell.h, minus the scale factors) ----
// forward: L = (x-1, y), M = (x-1, y-1), U = (x, y-1)
// SW: whether the model has short-gap switches (shortGapX <-> shortGapY).  None of the shipped models has (13 of cPecan's 15
// transitions); their two multiply-adds with a zero transition add exact zeros, so leaving them out changes no bit
// (the kernels are instantiated both ways, npr_batch_run picks by the loaded models).
template <bool SW = true>
__device__ __forceinline__ RCell rs_fwd_cell(const Trans &t, const RCell &L, const RCell &M, const RCell &U, float em, float exs, float exl,
                                             float eys, float eyl) {
    RCell c;
    float a;
    a = t.mm * M.m;
    a = __builtin_fmaf(t.sxm, M.sx, a);
    a = __builtin_fmaf(t.sym, M.sy, a);
    a = __builtin_fmaf(t.lxm, M.lx, a);
    a = __builtin_fmaf(t.lym, M.ly, a);
    c.m = em * a;
    a = t.msx * L.m;
    a = __builtin_fmaf(t.sxsx, L.sx, a);
    if constexpr (SW) a = __builtin_fmaf(t.sysx, L.sy, a);
    c.sx = exs * a;
    a = t.mlx * L.m;
    a = __builtin_fmaf(t.lxlx, L.lx, a);
    c.lx = exl * a;
    a = t.msy * U.m;
    a = __builtin_fmaf(t.sysy, U.sy, a);
    if constexpr (SW) a = __builtin_fmaf(t.sxsy, U.sx, a);
    c.sy = eys * a;
    a = t.mly * U.m;
    a = __builtin_fmaf(t.lyly, U.ly, a);
    c.ly = eyl * a;
    return c;
}
// backward: Ms = (x+1, y+1), Xs = (x+1, y), Ys = (x, y+1)
template <bool SW = true>
__device__ __forceinline__ RCell rs_bwd_cell(const Trans &t, const RCell &Ms, const RCell &Xs, const RCell &Ys, float em, float exs, float exl,
                                             float eys, float eyl) {
    const float am = em * Ms.m;
    const float asx = exs * Xs.sx;
    const float alx = exl * Xs.lx;
    const float asy = eys * Ys.sy;
    const float aly = eyl * Ys.ly;
    RCell c;
    float b;
    b = t.mm * am;
    b = __builtin_fmaf(t.msx, asx, b);
    b = __builtin_fmaf(t.mlx, alx, b);
    b = __builtin_fmaf(t.msy, asy, b);
    b = __builtin_fmaf(t.mly, aly, b);
    c.m = b;
    b = t.sxm * am;
    b = __builtin_fmaf(t.sxsx, asx, b);
    if constexpr (SW) b = __builtin_fmaf(t.sxsy, asy, b);
    c.sx = b;
    b = t.sym * am;
    b = __builtin_fmaf(t.sysy, asy, b);
    if constexpr (SW) b = __builtin_fmaf(t.sysx, asx, b);
    c.sy = b;
    b = t.lxm * am;
    b = __builtin_fmaf(t.lxlx, alx, b);
    c.lx = b;
    b = t.lym * am;
    b = __builtin_fmaf(t.lyly, aly, b);
    c.ly = b;
    return c;
}
__device__ __forceinline__ float rs_dot5(const float *w, const RCell &c) {
    float a = w[0] * c.m;
    a = __builtin_fmaf(w[1], c.sx, a);
    a = __builtin_fmaf(w[2], c.sy, a);
    a = __builtin_fmaf(w[3], c.lx, a);
    a = __builtin_fmaf(w[4], c.ly, a);
    return a;
}

// ---- renormalisation of the two held rows ----
__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    // row_shr:1, 2, 4, 8 inside the rows of 16 lanes, then row_bcast:15 / :31 across them: lane 63 ends with the maximum.  The DPP
    // operand sits on the maximum itself (a lane without a valid source reads 0, the identity); written as assembly because the
    // compiler expands update_dpp + max into three instructions per stage.  (s_nop: DPP read of a VGPR the previous VALU wrote.)
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
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}
__device__ __forceinline__ uint32_t rcell_max_bits(const RCell &c) {
    return umax3(umax3(static_cast<uint32_t>(fbits(c.m)), static_cast<uint32_t>(fbits(c.sx)), static_cast<uint32_t>(fbits(c.sy))),
                 static_cast<uint32_t>(fbits(c.lx)), static_cast<uint32_t>(fbits(c.ly)));
}
__device__ __forceinline__ void rcell_scale(RCell &c, float f) { c.m *= f, c.sx *= f, c.sy *= f, c.lx *= f, c.ly *= f; }
// Brings the largest value of the two rows into [2^(RS_TOP-1), 2^RS_TOP); returns what to add to the rows' exponent
// (wave-uniform).  Near the TOP of fp32's range, not at 1: what matters is how far BELOW its row's maximum a cell can lie and
// still be a normal number -- RS_TOP + 126 binary orders --, because a cell far below the maximum of its forward row can still
// carry posterior mass when its backward value is as far above the others (an alternative placement of a long indel).  Nothing
// in the recurrence grows by more than the sum of the transitions into a state (< 5) per anti-diagonal, so RS_K steps stay
// below 2^127 from RS_TOP = 85.  A row pair without a normal number (all zero: a dead band) is left alone.
constexpr int RS_TOP = NPR_RS_TOP;
template <int R>
__device__ __forceinline__ int rs_renorm(RDiag<R> &P, RDiag<R> &Q) {
    uint32_t u = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) u = umax3(u, rcell_max_bits(P.c[r]), rcell_max_bits(Q.c[r]));
    const uint32_t top = wave_max_u32(u);
    const int eb = static_cast<int>(top >> 23);
    if (eb == 0) return 0;
    const int k = uni(min(max(RS_TOP + 126 - eb, -126), 127));  // (scalar: the rows' exponent and everything derived from it stay on the scalar unit)
    float f;
    asm("v_mov_b32 %0, %1" : "=v"(f) : "s"((k + 127) << 23));  // 2^k, in a vector register: twenty multiplies with a scalar operand each cost more
#pragma unroll
    for (int r = 0; r < R; ++r) rcell_scale(P.c[r], f), rcell_scale(Q.c[r], f);
    return -k;
}

// ---- frame rebase: the whole register state moves by one slot, in place ----
// ONE asm statement per register group (the cells of the held rows; the base streams) with the test for "no rebase" inside it.  Written as C++ around per-register asm
// (npr_frame.h's way) the rebase is a branch, the moved registers are new values on one side of it, and the compiler pays
// for the join by copying 20-odd registers on the path WITHOUT a rebase, every anti-diagonal.  Inside one statement there
// is no join to pay for: the hot path costs two scalar instructions.  dir: +1 every slot takes its upper neighbour (the
// vacated top slot takes 0 / the injected base), -1 its lower neighbour, 0 nothing.  (s_nop: a DPP read of a VGPR written by
// the previous VALU instruction needs two wait states, which the compiler cannot see through inline assembly.)
// ---- generated by tools/gen_rs_rebase.py ----
__device__ __forceinline__ void rs_rebase_rows(RDiag<1> &P, RDiag<1> &Q, int dir) {
    asm volatile("s_cmp_lg_u32 %10, 0\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "2:\n\t"
                 ".subsection 1\n\t"
                 "9:\n\t"
                 "s_cmp_lt_i32 %10, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %1, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %2, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %4, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %5, %5 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %6, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %8, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %9, %9 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_branch 2b\n\t"
                 "1:\n\t"
                 "s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %8, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %9, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_branch 2b\n\t"
                 ".subsection 0"
                 : "+v"(P.c[0].m), "+v"(P.c[0].sx), "+v"(P.c[0].sy), "+v"(P.c[0].lx), "+v"(P.c[0].ly), "+v"(Q.c[0].m), "+v"(Q.c[0].sx), "+v"(Q.c[0].sy), "+v"(Q.c[0].lx), "+v"(Q.c[0].ly)
                 : "s"(dir)
                 : "scc");
}
__device__ __forceinline__ void rs_rebase_rows(RDiag<2> &P, RDiag<2> &Q, int dir) {
    asm volatile("s_cmp_lg_u32 %20, 0\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "2:\n\t"
                 ".subsection 1\n\t"
                 "9:\n\t"
                 "s_cmp_lt_i32 %20, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "v_swap_b32 %0, %1\n\tv_swap_b32 %2, %3\n\tv_swap_b32 %4, %5\n\tv_swap_b32 %6, %7\n\tv_swap_b32 %8, %9\n\tv_swap_b32 %10, %11\n\tv_swap_b32 %12, %13\n\tv_swap_b32 %14, %15\n\tv_swap_b32 %16, %17\n\tv_swap_b32 %18, %19\n\ts_nop 1\n\tv_mov_b32_dpp %1, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %5, %5 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %9, %9 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %11, %11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %13, %13 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %15, %15 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %17, %17 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %19, %19 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_branch 2b\n\t"
                 "1:\n\t"
                 "v_swap_b32 %1, %0\n\tv_swap_b32 %3, %2\n\tv_swap_b32 %5, %4\n\tv_swap_b32 %7, %6\n\tv_swap_b32 %9, %8\n\tv_swap_b32 %11, %10\n\tv_swap_b32 %13, %12\n\tv_swap_b32 %15, %14\n\tv_swap_b32 %17, %16\n\tv_swap_b32 %19, %18\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %8, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %10, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %12, %12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %14, %14 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %16, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %18, %18 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_branch 2b\n\t"
                 ".subsection 0"
                 : "+v"(P.c[0].m), "+v"(P.c[1].m), "+v"(P.c[0].sx), "+v"(P.c[1].sx), "+v"(P.c[0].sy), "+v"(P.c[1].sy), "+v"(P.c[0].lx), "+v"(P.c[1].lx), "+v"(P.c[0].ly), "+v"(P.c[1].ly), "+v"(Q.c[0].m), "+v"(Q.c[1].m), "+v"(Q.c[0].sx), "+v"(Q.c[1].sx), "+v"(Q.c[0].sy), "+v"(Q.c[1].sy), "+v"(Q.c[0].lx), "+v"(Q.c[1].lx), "+v"(Q.c[0].ly), "+v"(Q.c[1].ly)
                 : "s"(dir)
                 : "scc");
}
__device__ __forceinline__ void rs_rebase_rows(RDiag<4> &P, int dir) {
    asm volatile("s_cmp_lg_u32 %20, 0\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "2:\n\t"
                 ".subsection 1\n\t"
                 "9:\n\t"
                 "s_cmp_lt_i32 %20, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "v_swap_b32 %0, %1\n\tv_swap_b32 %1, %2\n\tv_swap_b32 %2, %3\n\tv_swap_b32 %4, %5\n\tv_swap_b32 %5, %6\n\tv_swap_b32 %6, %7\n\tv_swap_b32 %8, %9\n\tv_swap_b32 %9, %10\n\tv_swap_b32 %10, %11\n\tv_swap_b32 %12, %13\n\tv_swap_b32 %13, %14\n\tv_swap_b32 %14, %15\n\tv_swap_b32 %16, %17\n\tv_swap_b32 %17, %18\n\tv_swap_b32 %18, %19\n\ts_nop 1\n\tv_mov_b32_dpp %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %11, %11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %15, %15 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %19, %19 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_branch 2b\n\t"
                 "1:\n\t"
                 "v_swap_b32 %3, %2\n\tv_swap_b32 %2, %1\n\tv_swap_b32 %1, %0\n\tv_swap_b32 %7, %6\n\tv_swap_b32 %6, %5\n\tv_swap_b32 %5, %4\n\tv_swap_b32 %11, %10\n\tv_swap_b32 %10, %9\n\tv_swap_b32 %9, %8\n\tv_swap_b32 %15, %14\n\tv_swap_b32 %14, %13\n\tv_swap_b32 %13, %12\n\tv_swap_b32 %19, %18\n\tv_swap_b32 %18, %17\n\tv_swap_b32 %17, %16\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %8, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %12, %12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %16, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_branch 2b\n\t"
                 ".subsection 0"
                 : "+v"(P.c[0].m), "+v"(P.c[1].m), "+v"(P.c[2].m), "+v"(P.c[3].m), "+v"(P.c[0].sx), "+v"(P.c[1].sx), "+v"(P.c[2].sx), "+v"(P.c[3].sx), "+v"(P.c[0].sy), "+v"(P.c[1].sy), "+v"(P.c[2].sy), "+v"(P.c[3].sy), "+v"(P.c[0].lx), "+v"(P.c[1].lx), "+v"(P.c[2].lx), "+v"(P.c[3].lx), "+v"(P.c[0].ly), "+v"(P.c[1].ly), "+v"(P.c[2].ly), "+v"(P.c[3].ly)
                 : "s"(dir)
                 : "scc");
}
__device__ __forceinline__ void rs_rebase_rows(RDiag<4> &P, RDiag<4> &Q, int dir) { rs_rebase_rows(P, dir), rs_rebase_rows(Q, dir); }
__device__ __forceinline__ void rs_rebase_streams_fwd(Bases<1> &X, Bases<1> &Y, const Feed &fx, const Feed &fy, int dir, int offX, int offY, int xcap, int ycap) {
    int tmp;
    asm volatile("s_cmp_lg_u32 %3, 0\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "2:\n\t"
                 ".subsection 1\n\t"
                 "9:\n\t"
                 "s_cmp_lt_i32 %3, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %8, 64\n\ts_cbranch_scc0 31f\n\ts_nop 3\n\tv_readlane_b32 %2, %4, %8\n\ts_branch 41f\n\t31:\n\ts_sub_i32 %2, %8, 64\n\ts_nop 3\n\tv_readlane_b32 %2, %5, %2\n\t41:\n\ts_nop 3\n\tv_writelane_b32 %0, %2, 63\n\tv_writelane_b32 %1, %11, 63\n\t"
                 "s_branch 2b\n\t"
                 "1:\n\t"
                 "s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %9, 64\n\ts_cbranch_scc0 32f\n\ts_nop 3\n\tv_readlane_b32 %2, %6, %9\n\ts_branch 42f\n\t32:\n\ts_sub_i32 %2, %9, 64\n\ts_nop 3\n\tv_readlane_b32 %2, %7, %2\n\t42:\n\ts_nop 3\n\tv_writelane_b32 %1, %2, 0\n\tv_writelane_b32 %0, %10, 0\n\t"
                 "s_branch 2b\n\t"
                 ".subsection 0"
                 : "+v"(X.b[0]), "+v"(Y.b[0]), "=&s"(tmp)
                 : "s"(dir), "v"(fx.cur), "v"(fx.nxt), "v"(fy.cur), "v"(fy.nxt), "s"(offX), "s"(offY), "s"(xcap), "s"(ycap)
                 : "scc");
}
__device__ __forceinline__ void rs_rebase_streams_bwd(Bases<1> &X, Bases<1> &Y, const Feed &fx, const Feed &fy, int dir, int offX, int offY, int xcap, int ycap) {
    int tmp;
    asm volatile("s_cmp_lg_u32 %3, 0\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "2:\n\t"
                 ".subsection 1\n\t"
                 "9:\n\t"
                 "s_cmp_lt_i32 %3, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %9, 64\n\ts_cbranch_scc0 31f\n\ts_nop 3\n\tv_readlane_b32 %2, %6, %9\n\ts_branch 41f\n\t31:\n\ts_sub_i32 %2, %9, 64\n\ts_nop 3\n\tv_readlane_b32 %2, %7, %2\n\t41:\n\ts_nop 3\n\tv_writelane_b32 %1, %2, 63\n\tv_writelane_b32 %0, %10, 63\n\t"
                 "s_branch 2b\n\t"
                 "1:\n\t"
                 "s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %8, 64\n\ts_cbranch_scc0 32f\n\ts_nop 3\n\tv_readlane_b32 %2, %4, %8\n\ts_branch 42f\n\t32:\n\ts_sub_i32 %2, %8, 64\n\ts_nop 3\n\tv_readlane_b32 %2, %5, %2\n\t42:\n\ts_nop 3\n\tv_writelane_b32 %0, %2, 0\n\tv_writelane_b32 %1, %11, 0\n\t"
                 "s_branch 2b\n\t"
                 ".subsection 0"
                 : "+v"(X.b[0]), "+v"(Y.b[0]), "=&s"(tmp)
                 : "s"(dir), "v"(fx.cur), "v"(fx.nxt), "v"(fy.cur), "v"(fy.nxt), "s"(offX), "s"(offY), "s"(xcap), "s"(ycap)
                 : "scc");
}
__device__ __forceinline__ void rs_rebase_streams_fwd(Bases<2> &X, Bases<2> &Y, const Feed &fx, const Feed &fy, int dir, int offX, int offY, int xcap, int ycap) {
    int tmp;
    asm volatile("s_cmp_lg_u32 %5, 0\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "2:\n\t"
                 ".subsection 1\n\t"
                 "9:\n\t"
                 "s_cmp_lt_i32 %5, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "v_swap_b32 %0, %1\n\tv_swap_b32 %2, %3\n\ts_nop 1\n\tv_mov_b32_dpp %1, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %10, 64\n\ts_cbranch_scc0 31f\n\ts_nop 3\n\tv_readlane_b32 %4, %6, %10\n\ts_branch 41f\n\t31:\n\ts_sub_i32 %4, %10, 64\n\ts_nop 3\n\tv_readlane_b32 %4, %7, %4\n\t41:\n\ts_nop 3\n\tv_writelane_b32 %1, %4, 63\n\tv_writelane_b32 %3, %13, 63\n\t"
                 "s_branch 2b\n\t"
                 "1:\n\t"
                 "v_swap_b32 %1, %0\n\tv_swap_b32 %3, %2\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %11, 64\n\ts_cbranch_scc0 32f\n\ts_nop 3\n\tv_readlane_b32 %4, %8, %11\n\ts_branch 42f\n\t32:\n\ts_sub_i32 %4, %11, 64\n\ts_nop 3\n\tv_readlane_b32 %4, %9, %4\n\t42:\n\ts_nop 3\n\tv_writelane_b32 %2, %4, 0\n\tv_writelane_b32 %0, %12, 0\n\t"
                 "s_branch 2b\n\t"
                 ".subsection 0"
                 : "+v"(X.b[0]), "+v"(X.b[1]), "+v"(Y.b[0]), "+v"(Y.b[1]), "=&s"(tmp)
                 : "s"(dir), "v"(fx.cur), "v"(fx.nxt), "v"(fy.cur), "v"(fy.nxt), "s"(offX), "s"(offY), "s"(xcap), "s"(ycap)
                 : "scc");
}
__device__ __forceinline__ void rs_rebase_streams_bwd(Bases<2> &X, Bases<2> &Y, const Feed &fx, const Feed &fy, int dir, int offX, int offY, int xcap, int ycap) {
    int tmp;
    asm volatile("s_cmp_lg_u32 %5, 0\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "2:\n\t"
                 ".subsection 1\n\t"
                 "9:\n\t"
                 "s_cmp_lt_i32 %5, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "v_swap_b32 %0, %1\n\tv_swap_b32 %2, %3\n\ts_nop 1\n\tv_mov_b32_dpp %1, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %11, 64\n\ts_cbranch_scc0 31f\n\ts_nop 3\n\tv_readlane_b32 %4, %8, %11\n\ts_branch 41f\n\t31:\n\ts_sub_i32 %4, %11, 64\n\ts_nop 3\n\tv_readlane_b32 %4, %9, %4\n\t41:\n\ts_nop 3\n\tv_writelane_b32 %3, %4, 63\n\tv_writelane_b32 %1, %12, 63\n\t"
                 "s_branch 2b\n\t"
                 "1:\n\t"
                 "v_swap_b32 %1, %0\n\tv_swap_b32 %3, %2\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %10, 64\n\ts_cbranch_scc0 32f\n\ts_nop 3\n\tv_readlane_b32 %4, %6, %10\n\ts_branch 42f\n\t32:\n\ts_sub_i32 %4, %10, 64\n\ts_nop 3\n\tv_readlane_b32 %4, %7, %4\n\t42:\n\ts_nop 3\n\tv_writelane_b32 %0, %4, 0\n\tv_writelane_b32 %2, %13, 0\n\t"
                 "s_branch 2b\n\t"
                 ".subsection 0"
                 : "+v"(X.b[0]), "+v"(X.b[1]), "+v"(Y.b[0]), "+v"(Y.b[1]), "=&s"(tmp)
                 : "s"(dir), "v"(fx.cur), "v"(fx.nxt), "v"(fy.cur), "v"(fy.nxt), "s"(offX), "s"(offY), "s"(xcap), "s"(ycap)
                 : "scc");
}
__device__ __forceinline__ void rs_rebase_streams_fwd(Bases<4> &X, Bases<4> &Y, const Feed &fx, const Feed &fy, int dir, int offX, int offY, int xcap, int ycap) {
    int tmp;
    asm volatile("s_cmp_lg_u32 %9, 0\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "2:\n\t"
                 ".subsection 1\n\t"
                 "9:\n\t"
                 "s_cmp_lt_i32 %9, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "v_swap_b32 %0, %1\n\tv_swap_b32 %1, %2\n\tv_swap_b32 %2, %3\n\tv_swap_b32 %4, %5\n\tv_swap_b32 %5, %6\n\tv_swap_b32 %6, %7\n\ts_nop 1\n\tv_mov_b32_dpp %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %14, 64\n\ts_cbranch_scc0 31f\n\ts_nop 3\n\tv_readlane_b32 %8, %10, %14\n\ts_branch 41f\n\t31:\n\ts_sub_i32 %8, %14, 64\n\ts_nop 3\n\tv_readlane_b32 %8, %11, %8\n\t41:\n\ts_nop 3\n\tv_writelane_b32 %3, %8, 63\n\tv_writelane_b32 %7, %17, 63\n\t"
                 "s_branch 2b\n\t"
                 "1:\n\t"
                 "v_swap_b32 %3, %2\n\tv_swap_b32 %2, %1\n\tv_swap_b32 %1, %0\n\tv_swap_b32 %7, %6\n\tv_swap_b32 %6, %5\n\tv_swap_b32 %5, %4\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %15, 64\n\ts_cbranch_scc0 32f\n\ts_nop 3\n\tv_readlane_b32 %8, %12, %15\n\ts_branch 42f\n\t32:\n\ts_sub_i32 %8, %15, 64\n\ts_nop 3\n\tv_readlane_b32 %8, %13, %8\n\t42:\n\ts_nop 3\n\tv_writelane_b32 %4, %8, 0\n\tv_writelane_b32 %0, %16, 0\n\t"
                 "s_branch 2b\n\t"
                 ".subsection 0"
                 : "+v"(X.b[0]), "+v"(X.b[1]), "+v"(X.b[2]), "+v"(X.b[3]), "+v"(Y.b[0]), "+v"(Y.b[1]), "+v"(Y.b[2]), "+v"(Y.b[3]), "=&s"(tmp)
                 : "s"(dir), "v"(fx.cur), "v"(fx.nxt), "v"(fy.cur), "v"(fy.nxt), "s"(offX), "s"(offY), "s"(xcap), "s"(ycap)
                 : "scc");
}
__device__ __forceinline__ void rs_rebase_streams_bwd(Bases<4> &X, Bases<4> &Y, const Feed &fx, const Feed &fy, int dir, int offX, int offY, int xcap, int ycap) {
    int tmp;
    asm volatile("s_cmp_lg_u32 %9, 0\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "2:\n\t"
                 ".subsection 1\n\t"
                 "9:\n\t"
                 "s_cmp_lt_i32 %9, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "v_swap_b32 %0, %1\n\tv_swap_b32 %1, %2\n\tv_swap_b32 %2, %3\n\tv_swap_b32 %4, %5\n\tv_swap_b32 %5, %6\n\tv_swap_b32 %6, %7\n\ts_nop 1\n\tv_mov_b32_dpp %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %15, 64\n\ts_cbranch_scc0 31f\n\ts_nop 3\n\tv_readlane_b32 %8, %12, %15\n\ts_branch 41f\n\t31:\n\ts_sub_i32 %8, %15, 64\n\ts_nop 3\n\tv_readlane_b32 %8, %13, %8\n\t41:\n\ts_nop 3\n\tv_writelane_b32 %7, %8, 63\n\tv_writelane_b32 %3, %16, 63\n\t"
                 "s_branch 2b\n\t"
                 "1:\n\t"
                 "v_swap_b32 %3, %2\n\tv_swap_b32 %2, %1\n\tv_swap_b32 %1, %0\n\tv_swap_b32 %7, %6\n\tv_swap_b32 %6, %5\n\tv_swap_b32 %5, %4\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %14, 64\n\ts_cbranch_scc0 32f\n\ts_nop 3\n\tv_readlane_b32 %8, %10, %14\n\ts_branch 42f\n\t32:\n\ts_sub_i32 %8, %14, 64\n\ts_nop 3\n\tv_readlane_b32 %8, %11, %8\n\t42:\n\ts_nop 3\n\tv_writelane_b32 %0, %8, 0\n\tv_writelane_b32 %4, %17, 0\n\t"
                 "s_branch 2b\n\t"
                 ".subsection 0"
                 : "+v"(X.b[0]), "+v"(X.b[1]), "+v"(X.b[2]), "+v"(X.b[3]), "+v"(Y.b[0]), "+v"(Y.b[1]), "+v"(Y.b[2]), "+v"(Y.b[3]), "=&s"(tmp)
                 : "s"(dir), "v"(fx.cur), "v"(fx.nxt), "v"(fy.cur), "v"(fy.nxt), "s"(offX), "s"(offY), "s"(xcap), "s"(ycap)
                 : "scc");
}
__device__ __forceinline__ void rs_rebase_all_fwd(RDiag<1> &P, RDiag<1> &Q, Bases<1> &X, Bases<1> &Y, const Feed &fx, const Feed &fy, int dir, int offX, int offY, int xcap, int ycap) {
    int tmp;
    asm volatile("s_cmp_lg_u32 %13, 0\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "2:\n\t"
                 ".subsection 1\n\t"
                 "9:\n\t"
                 "s_cmp_lt_i32 %13, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %1, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %2, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %4, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %5, %5 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %6, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %8, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %9, %9 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %10, %10 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %11, %11 wave_shl:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %18, 64\n\ts_cbranch_scc0 31f\n\ts_nop 3\n\tv_readlane_b32 %12, %14, %18\n\ts_branch 41f\n\t31:\n\ts_sub_i32 %12, %18, 64\n\ts_nop 3\n\tv_readlane_b32 %12, %15, %12\n\t41:\n\ts_nop 3\n\tv_writelane_b32 %10, %12, 63\n\tv_writelane_b32 %11, %21, 63\n\t"
                 "s_branch 2b\n\t"
                 "1:\n\t"
                 "s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %8, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %9, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %10, %10 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %11, %11 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %19, 64\n\ts_cbranch_scc0 32f\n\ts_nop 3\n\tv_readlane_b32 %12, %16, %19\n\ts_branch 42f\n\t32:\n\ts_sub_i32 %12, %19, 64\n\ts_nop 3\n\tv_readlane_b32 %12, %17, %12\n\t42:\n\ts_nop 3\n\tv_writelane_b32 %11, %12, 0\n\tv_writelane_b32 %10, %20, 0\n\t"
                 "s_branch 2b\n\t"
                 ".subsection 0"
                 : "+v"(P.c[0].m), "+v"(P.c[0].sx), "+v"(P.c[0].sy), "+v"(P.c[0].lx), "+v"(P.c[0].ly), "+v"(Q.c[0].m), "+v"(Q.c[0].sx), "+v"(Q.c[0].sy), "+v"(Q.c[0].lx), "+v"(Q.c[0].ly), "+v"(X.b[0]), "+v"(Y.b[0]), "=&s"(tmp)
                 : "s"(dir), "v"(fx.cur), "v"(fx.nxt), "v"(fy.cur), "v"(fy.nxt), "s"(offX), "s"(offY), "s"(xcap), "s"(ycap)
                 : "scc");
}
__device__ __forceinline__ void rs_rebase_all_bwd(RDiag<1> &P, RDiag<1> &Q, Bases<1> &X, Bases<1> &Y, const Feed &fx, const Feed &fy, int dir, int offX, int offY, int xcap, int ycap) {
    int tmp;
    asm volatile("s_cmp_lg_u32 %13, 0\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "2:\n\t"
                 ".subsection 1\n\t"
                 "9:\n\t"
                 "s_cmp_lt_i32 %13, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %1, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %2, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %4, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %5, %5 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %6, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %8, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %9, %9 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %10, %10 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %11, %11 wave_shl:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %19, 64\n\ts_cbranch_scc0 31f\n\ts_nop 3\n\tv_readlane_b32 %12, %16, %19\n\ts_branch 41f\n\t31:\n\ts_sub_i32 %12, %19, 64\n\ts_nop 3\n\tv_readlane_b32 %12, %17, %12\n\t41:\n\ts_nop 3\n\tv_writelane_b32 %11, %12, 63\n\tv_writelane_b32 %10, %20, 63\n\t"
                 "s_branch 2b\n\t"
                 "1:\n\t"
                 "s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %8, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %9, %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %10, %10 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %11, %11 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %18, 64\n\ts_cbranch_scc0 32f\n\ts_nop 3\n\tv_readlane_b32 %12, %14, %18\n\ts_branch 42f\n\t32:\n\ts_sub_i32 %12, %18, 64\n\ts_nop 3\n\tv_readlane_b32 %12, %15, %12\n\t42:\n\ts_nop 3\n\tv_writelane_b32 %10, %12, 0\n\tv_writelane_b32 %11, %21, 0\n\t"
                 "s_branch 2b\n\t"
                 ".subsection 0"
                 : "+v"(P.c[0].m), "+v"(P.c[0].sx), "+v"(P.c[0].sy), "+v"(P.c[0].lx), "+v"(P.c[0].ly), "+v"(Q.c[0].m), "+v"(Q.c[0].sx), "+v"(Q.c[0].sy), "+v"(Q.c[0].lx), "+v"(Q.c[0].ly), "+v"(X.b[0]), "+v"(Y.b[0]), "=&s"(tmp)
                 : "s"(dir), "v"(fx.cur), "v"(fx.nxt), "v"(fy.cur), "v"(fy.nxt), "s"(offX), "s"(offY), "s"(xcap), "s"(ycap)
                 : "scc");
}
__device__ __forceinline__ void rs_rebase_all_fwd(RDiag<2> &P, RDiag<2> &Q, Bases<2> &X, Bases<2> &Y, const Feed &fx, const Feed &fy, int dir, int offX, int offY, int xcap, int ycap) {
    int tmp;
    asm volatile("s_cmp_lg_u32 %25, 0\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "2:\n\t"
                 ".subsection 1\n\t"
                 "9:\n\t"
                 "s_cmp_lt_i32 %25, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "v_swap_b32 %0, %1\n\tv_swap_b32 %2, %3\n\tv_swap_b32 %4, %5\n\tv_swap_b32 %6, %7\n\tv_swap_b32 %8, %9\n\tv_swap_b32 %10, %11\n\tv_swap_b32 %12, %13\n\tv_swap_b32 %14, %15\n\tv_swap_b32 %16, %17\n\tv_swap_b32 %18, %19\n\tv_swap_b32 %20, %21\n\tv_swap_b32 %22, %23\n\ts_nop 1\n\tv_mov_b32_dpp %1, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %5, %5 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %9, %9 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %11, %11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %13, %13 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %15, %15 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %17, %17 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %19, %19 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %21, %21 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %23, %23 wave_shl:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %30, 64\n\ts_cbranch_scc0 31f\n\ts_nop 3\n\tv_readlane_b32 %24, %26, %30\n\ts_branch 41f\n\t31:\n\ts_sub_i32 %24, %30, 64\n\ts_nop 3\n\tv_readlane_b32 %24, %27, %24\n\t41:\n\ts_nop 3\n\tv_writelane_b32 %21, %24, 63\n\tv_writelane_b32 %23, %33, 63\n\t"
                 "s_branch 2b\n\t"
                 "1:\n\t"
                 "v_swap_b32 %1, %0\n\tv_swap_b32 %3, %2\n\tv_swap_b32 %5, %4\n\tv_swap_b32 %7, %6\n\tv_swap_b32 %9, %8\n\tv_swap_b32 %11, %10\n\tv_swap_b32 %13, %12\n\tv_swap_b32 %15, %14\n\tv_swap_b32 %17, %16\n\tv_swap_b32 %19, %18\n\tv_swap_b32 %21, %20\n\tv_swap_b32 %23, %22\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %8, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %10, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %12, %12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %14, %14 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %16, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %18, %18 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %20, %20 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %22, %22 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %31, 64\n\ts_cbranch_scc0 32f\n\ts_nop 3\n\tv_readlane_b32 %24, %28, %31\n\ts_branch 42f\n\t32:\n\ts_sub_i32 %24, %31, 64\n\ts_nop 3\n\tv_readlane_b32 %24, %29, %24\n\t42:\n\ts_nop 3\n\tv_writelane_b32 %22, %24, 0\n\tv_writelane_b32 %20, %32, 0\n\t"
                 "s_branch 2b\n\t"
                 ".subsection 0"
                 : "+v"(P.c[0].m), "+v"(P.c[1].m), "+v"(P.c[0].sx), "+v"(P.c[1].sx), "+v"(P.c[0].sy), "+v"(P.c[1].sy), "+v"(P.c[0].lx), "+v"(P.c[1].lx), "+v"(P.c[0].ly), "+v"(P.c[1].ly), "+v"(Q.c[0].m), "+v"(Q.c[1].m), "+v"(Q.c[0].sx), "+v"(Q.c[1].sx), "+v"(Q.c[0].sy), "+v"(Q.c[1].sy), "+v"(Q.c[0].lx), "+v"(Q.c[1].lx), "+v"(Q.c[0].ly), "+v"(Q.c[1].ly), "+v"(X.b[0]), "+v"(X.b[1]), "+v"(Y.b[0]), "+v"(Y.b[1]), "=&s"(tmp)
                 : "s"(dir), "v"(fx.cur), "v"(fx.nxt), "v"(fy.cur), "v"(fy.nxt), "s"(offX), "s"(offY), "s"(xcap), "s"(ycap)
                 : "scc");
}
__device__ __forceinline__ void rs_rebase_all_bwd(RDiag<2> &P, RDiag<2> &Q, Bases<2> &X, Bases<2> &Y, const Feed &fx, const Feed &fy, int dir, int offX, int offY, int xcap, int ycap) {
    int tmp;
    asm volatile("s_cmp_lg_u32 %25, 0\n\t"
                 "s_cbranch_scc1 9f\n\t"
                 "2:\n\t"
                 ".subsection 1\n\t"
                 "9:\n\t"
                 "s_cmp_lt_i32 %25, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "v_swap_b32 %0, %1\n\tv_swap_b32 %2, %3\n\tv_swap_b32 %4, %5\n\tv_swap_b32 %6, %7\n\tv_swap_b32 %8, %9\n\tv_swap_b32 %10, %11\n\tv_swap_b32 %12, %13\n\tv_swap_b32 %14, %15\n\tv_swap_b32 %16, %17\n\tv_swap_b32 %18, %19\n\tv_swap_b32 %20, %21\n\tv_swap_b32 %22, %23\n\ts_nop 1\n\tv_mov_b32_dpp %1, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %5, %5 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %7, %7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %9, %9 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %11, %11 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %13, %13 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %15, %15 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %17, %17 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %19, %19 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %21, %21 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %23, %23 wave_shl:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %31, 64\n\ts_cbranch_scc0 31f\n\ts_nop 3\n\tv_readlane_b32 %24, %28, %31\n\ts_branch 41f\n\t31:\n\ts_sub_i32 %24, %31, 64\n\ts_nop 3\n\tv_readlane_b32 %24, %29, %24\n\t41:\n\ts_nop 3\n\tv_writelane_b32 %23, %24, 63\n\tv_writelane_b32 %21, %32, 63\n\t"
                 "s_branch 2b\n\t"
                 "1:\n\t"
                 "v_swap_b32 %1, %0\n\tv_swap_b32 %3, %2\n\tv_swap_b32 %5, %4\n\tv_swap_b32 %7, %6\n\tv_swap_b32 %9, %8\n\tv_swap_b32 %11, %10\n\tv_swap_b32 %13, %12\n\tv_swap_b32 %15, %14\n\tv_swap_b32 %17, %16\n\tv_swap_b32 %19, %18\n\tv_swap_b32 %21, %20\n\tv_swap_b32 %23, %22\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %8, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %10, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %12, %12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %14, %14 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %16, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %18, %18 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\tv_mov_b32_dpp %20, %20 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %22, %22 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_cmp_lt_i32 %30, 64\n\ts_cbranch_scc0 32f\n\ts_nop 3\n\tv_readlane_b32 %24, %26, %30\n\ts_branch 42f\n\t32:\n\ts_sub_i32 %24, %30, 64\n\ts_nop 3\n\tv_readlane_b32 %24, %27, %24\n\t42:\n\ts_nop 3\n\tv_writelane_b32 %20, %24, 0\n\tv_writelane_b32 %22, %33, 0\n\t"
                 "s_branch 2b\n\t"
                 ".subsection 0"
                 : "+v"(P.c[0].m), "+v"(P.c[1].m), "+v"(P.c[0].sx), "+v"(P.c[1].sx), "+v"(P.c[0].sy), "+v"(P.c[1].sy), "+v"(P.c[0].lx), "+v"(P.c[1].lx), "+v"(P.c[0].ly), "+v"(P.c[1].ly), "+v"(Q.c[0].m), "+v"(Q.c[1].m), "+v"(Q.c[0].sx), "+v"(Q.c[1].sx), "+v"(Q.c[0].sy), "+v"(Q.c[1].sy), "+v"(Q.c[0].lx), "+v"(Q.c[1].lx), "+v"(Q.c[0].ly), "+v"(Q.c[1].ly), "+v"(X.b[0]), "+v"(X.b[1]), "+v"(Y.b[0]), "+v"(Y.b[1]), "=&s"(tmp)
                 : "s"(dir), "v"(fx.cur), "v"(fx.nxt), "v"(fy.cur), "v"(fy.nxt), "s"(offX), "s"(offY), "s"(xcap), "s"(ycap)
                 : "scc");
}
// ---- end of generated code ----

// ... and the scalar side of it: the frame's origin moves by one lattice point
__device__ __forceinline__ void rs_rebase_origin(int &x0, int &y0, int dir) {
    int xs = uni(x0), ys = uni(y0);
    asm volatile("s_add_i32 %0, %0, %2\n\t"
                 "s_sub_i32 %1, %1, %2"
                 : "+s"(xs), "+s"(ys)
                 : "s"(dir)
                 : "scc");
    x0 = xs, y0 = ys;
}

#ifndef NPR_RS_ONE_REBASE_MAX_R
#define NPR_RS_ONE_REBASE_MAX_R 2  // slots per lane up to which rows and streams rebase in ONE asm statement (rs_rebase_all_*)
#endif
// A sweep's register state: the even anti-diagonals in A, the odd ones in B, the base streams and the rows' common exponent.
template <int R>
struct RsState {
    RDiag<R> A, B;
    Streams<R> S;
    int x0, y0;
    int e;
};

// Frame rebase of the forward sweep, r = +1: (x0, y0) -> (x0 + 1, y0 - 1), every slot takes its upper neighbour; r = 0: nothing.
// Called on every anti-diagonal; nothing here is conditional in C++: the statements test r themselves, and the base a stream
// takes in from its feed is read inside the statement (from the feed's current or next block: at most one base ahead of
// the steps, which do the refills).
template <int R>
__device__ __forceinline__ void rs_fwd_rebase(const StepEnv &E, int r, RsState<R> &Q) {
    const int dir = uni(r);
    const int offX = uni((Q.x0 + 64 * R - 1) - Q.S.fx.base);  // up: the X stream takes in X[(x0 + 1) + 64R - 2]
    const int offY = uni(Q.y0 - Q.S.fy.base);                  // down: the Y stream takes in Y[(y0 + 1) - 1]
    if constexpr (R <= NPR_RS_ONE_REBASE_MAX_R) {  // rows and streams in one statement: one skip test per anti-diagonal
        rs_rebase_all_fwd(Q.A, Q.B, Q.S.X, Q.S.Y, Q.S.fx, Q.S.fy, dir, offX, offY, uni(Q.S.xcap), uni(Q.S.ycap));
        rs_rebase_origin(Q.x0, Q.y0, dir);
    } else {
        rs_rebase_streams_fwd(Q.S.X, Q.S.Y, Q.S.fx, Q.S.fy, dir, offX, offY, uni(Q.S.xcap), uni(Q.S.ycap));
        rs_rebase_origin(Q.x0, Q.y0, dir);
        rs_rebase_rows(Q.A, Q.B, dir);
    }
}
// ... and of the backward sweep, which undoes the forward one: r is the forward rebase being undone.
template <int R>
__device__ __forceinline__ void rs_bwd_rebase(const StepEnv &E, int r, RsState<R> &Q) {
    const int dir = uni(-r);
    const int offX = uni(Q.S.fx.base - (Q.x0 - 1));        // down (r > 0): the X stream takes in X[x0 - 1] at slot 0
    const int offY = uni(Q.S.fy.base - (Q.y0 - 64 * R));   // up (r < 0): the Y stream takes in Y[(y0 - 1) - (64R - 1)] on top
    if constexpr (R <= NPR_RS_ONE_REBASE_MAX_R) {
        rs_rebase_all_bwd(Q.A, Q.B, Q.S.X, Q.S.Y, Q.S.fx, Q.S.fy, dir, offX, offY, uni(Q.S.xcap), uni(Q.S.ycap));
        rs_rebase_origin(Q.x0, Q.y0, dir);
    } else {
        rs_rebase_streams_bwd(Q.S.X, Q.S.Y, Q.S.fx, Q.S.fy, dir, offX, offY, uni(Q.S.xcap), uni(Q.S.ycap));
        rs_rebase_origin(Q.x0, Q.y0, dir);
        rs_rebase_rows(Q.A, Q.B, dir);
    }
}

// The new row replaces the one two anti-diagonals away, under the band's lane mask (the arithmetic itself runs under the
// mask: no select per value) ...
template <class F>
__device__ __forceinline__ void rs_put(RCell &dst, uint64_t in_band, F &&cell) {
    if (__builtin_expect(lanes_of(in_band), 1)) dst = cell();  // (expected: keeps the block in line instead of behind two taken branches)
}
// ... and when the band is not where it was two anti-diagonals ago (`moved`: a bit of the control word, npr_sched.h; once
// in ten anti-diagonals on noisy guides) everything outside it is cleared: the row that was overwritten may have had cells
// there.  One scalar test per anti-diagonal; no masks of the held rows to carry along.
template <int R>
__device__ __forceinline__ void rs_clear_outside(RDiag<R> &io, const Masks<R> &mk, uint32_t moved) {
    if (moved) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (lanes_of(~mk.cell[r])) io.c[r] = zero_rcell();
    }
}

// One forward anti-diagonal: `io` holds d-2 on entry and d on exit, `p1` holds d-1.  S.X / S.Y: X[x-1]*8, Y[y-1]*8.
template <int R, bool CHK = true, bool SW = true, bool FLAT = false>
__device__ __forceinline__ void rs_fwd_x_step(const StepEnv &E, RDiag<R> &io, const RDiag<R> &p1, Streams<R> &S, int &x0,
                                              const Masks<R> &mk, uint32_t moved) {
    x0 += 1;
    S.xcap = __builtin_amdgcn_readlane(S.X.b[0], 0);
    bases_up<R>(S.X, feed8_take<+1, RS_XS, CHK>(S.fx, E.X, E.lX, x0 + 64 * R - 2, E.lane));
    const RDiag<R> U = rs_shift_up<R>(p1);  // (x, y-1) is slot j+1 of d-1; (x-1, y) keeps slot j
    RDiag<R> o;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float em, exs, exl, eys, eyl;
        if constexpr (R <= NPR_RS_DEADCODE_MAX_R) {
            rs_cell_emissions<R, FLAT>(E.ltab, mk.cell[r], S.X.b[r], S.Y.b[r], em, exs, exl, eys, eyl);
            o.c[r] = rs_fwd_cell<SW>(E.tr, p1.c[r], io.c[r], U.c[r], em, exs, exl, eys, eyl);
        } else {
            rs_cell_emissions<R, FLAT>(E.ltab, mk.cell[r], S.X.b[r], S.Y.b[r], em, exs, exl, eys, eyl);
            rs_put(io.c[r], mk.cell[r], [&] { return rs_fwd_cell<SW>(E.tr, p1.c[r], io.c[r], U.c[r], em, exs, exl, eys, eyl); });
        }
    }
    if constexpr (R <= NPR_RS_DEADCODE_MAX_R) io = o;
    else rs_clear_outside<R>(io, mk, moved);
}
template <int R, bool CHK = true, bool SW = true, bool FLAT = false>
__device__ __forceinline__ void rs_fwd_y_step(const StepEnv &E, RDiag<R> &io, const RDiag<R> &p1, Streams<R> &S, int &y0,
                                              const Masks<R> &mk, uint32_t moved) {
    y0 += 1;
    S.ycap = __builtin_amdgcn_readlane(S.Y.b[R - 1], 63);
    bases_down<R>(S.Y, feed8_take<+1, RS_YS, CHK>(S.fy, E.Y, E.lY, y0 - 1, E.lane));
    const RDiag<R> L = rs_shift_down<R>(p1);  // (x-1, y) is slot j-1 of d-1; (x, y-1) keeps slot j
    RDiag<R> o;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float em, exs, exl, eys, eyl;
        if constexpr (R <= NPR_RS_DEADCODE_MAX_R) {
            rs_cell_emissions<R, FLAT>(E.ltab, mk.cell[r], S.X.b[r], S.Y.b[r], em, exs, exl, eys, eyl);
            o.c[r] = rs_fwd_cell<SW>(E.tr, L.c[r], io.c[r], p1.c[r], em, exs, exl, eys, eyl);
        } else {
            rs_cell_emissions<R, FLAT>(E.ltab, mk.cell[r], S.X.b[r], S.Y.b[r], em, exs, exl, eys, eyl);
            rs_put(io.c[r], mk.cell[r], [&] { return rs_fwd_cell<SW>(E.tr, L.c[r], io.c[r], p1.c[r], em, exs, exl, eys, eyl); });
        }
    }
    if constexpr (R <= NPR_RS_DEADCODE_MAX_R) io = o;
    else rs_clear_outside<R>(io, mk, moved);
}
// One backward anti-diagonal d: `io` holds d+2 on entry and d on exit, `s1` holds d+1.  S.X / S.Y: X[x]*8, Y[y]*8.
template <int R, bool CHK = true, bool SW = true, bool FLAT = false>
__device__ __forceinline__ void rs_bwd_x_step(const StepEnv &E, RDiag<R> &io, const RDiag<R> &s1, Streams<R> &S, int &x0,
                                              const Masks<R> &mk, uint32_t moved) {
    x0 -= 1;
    S.xcap = __builtin_amdgcn_readlane(S.X.b[R - 1], 63);
    bases_down<R>(S.X, feed8_take<-1, RS_XS, CHK>(S.fx, E.X, E.lX, x0, E.lane));
    const RDiag<R> Ys = rs_shift_down<R>(s1);  // (x, y+1) is slot j-1 of d+1; (x+1, y) keeps slot j
    RDiag<R> o;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float em, exs, exl, eys, eyl;
        if constexpr (R <= NPR_RS_DEADCODE_MAX_R) {
            rs_cell_emissions<R, FLAT>(E.ltab, mk.cell[r], S.X.b[r], S.Y.b[r], em, exs, exl, eys, eyl);
            o.c[r] = rs_bwd_cell<SW>(E.tr, io.c[r], s1.c[r], Ys.c[r], em, exs, exl, eys, eyl);
        } else {
            rs_cell_emissions<R, FLAT>(E.ltab, mk.cell[r], S.X.b[r], S.Y.b[r], em, exs, exl, eys, eyl);
            rs_put(io.c[r], mk.cell[r], [&] { return rs_bwd_cell<SW>(E.tr, io.c[r], s1.c[r], Ys.c[r], em, exs, exl, eys, eyl); });
        }
    }
    if constexpr (R <= NPR_RS_DEADCODE_MAX_R) io = o;
    else rs_clear_outside<R>(io, mk, moved);
}
template <int R, bool CHK = true, bool SW = true, bool FLAT = false>
__device__ __forceinline__ void rs_bwd_y_step(const StepEnv &E, RDiag<R> &io, const RDiag<R> &s1, Streams<R> &S, int &y0,
                                              const Masks<R> &mk, uint32_t moved) {
    y0 -= 1;
    S.ycap = __builtin_amdgcn_readlane(S.Y.b[0], 0);
    bases_up<R>(S.Y, feed8_take<-1, RS_YS, CHK>(S.fy, E.Y, E.lY, y0 - (64 * R - 1), E.lane));
    const RDiag<R> Xs = rs_shift_up<R>(s1);  // (x+1, y) is slot j+1 of d+1; (x, y+1) keeps slot j
    RDiag<R> o;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float em, exs, exl, eys, eyl;
        if constexpr (R <= NPR_RS_DEADCODE_MAX_R) {
            rs_cell_emissions<R, FLAT>(E.ltab, mk.cell[r], S.X.b[r], S.Y.b[r], em, exs, exl, eys, eyl);
            o.c[r] = rs_bwd_cell<SW>(E.tr, io.c[r], Xs.c[r], s1.c[r], em, exs, exl, eys, eyl);
        } else {
            rs_cell_emissions<R, FLAT>(E.ltab, mk.cell[r], S.X.b[r], S.Y.b[r], em, exs, exl, eys, eyl);
            rs_put(io.c[r], mk.cell[r], [&] { return rs_bwd_cell<SW>(E.tr, io.c[r], Xs.c[r], s1.c[r], em, exs, exl, eys, eyl); });
        }
    }
    if constexpr (R <= NPR_RS_DEADCODE_MAX_R) io = o;
    else rs_clear_outside<R>(io, mk, moved);
}

// ---- forward rows in HBM: 4 bytes per slot (the match value); the row offsets of the control words are the 8-byte
// layout's (npr_sched.h), halved ----
typedef int v2i_rs __attribute__((ext_vector_type(2)));
template <int R>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rs_task_rsrc(char *F) {
    return __builtin_amdgcn_make_buffer_rsrc(F - static_cast<int64_t>(row_bias<R>() / 2), 0, -1, 0x00020000);
}
// cache policy of the forward rows' stores and loads.  Non-temporal (2), which pays in the stripe kernels (npr_kernel_tile_cs.hip), costs here: the headline
// launch 134.7 / 134.9 -> 139.8 / 139.7 ms, alternating runs (the stores alone 134.1 -> 137.6 / 138.6, the loads alone 136.7 / 136.7) -- a frame kernel's rows are
// read back by the partner sweep from the L2 they were written to.
#ifndef NPR_RS_ROW_AUX
#define NPR_RS_ROW_AUX 0
#endif
#ifndef NPR_RS_PAIR_NT
#define NPR_RS_PAIR_NT 0  // the posterior triples as non-temporal stores (read once, by the finish): nothing on the headline launch (132.6 / 132.6 ms without, 132.9 / 132.9 with)
#endif
#ifndef NPR_RS_ROW_ST_AUX
#define NPR_RS_ROW_ST_AUX NPR_RS_ROW_AUX
#endif
#ifndef NPR_RS_ROW_LD_AUX
#define NPR_RS_ROW_LD_AUX NPR_RS_ROW_AUX
#endif
constexpr int RS_ROW_ST_AUX = NPR_RS_ROW_ST_AUX, RS_ROW_LD_AUX = NPR_RS_ROW_LD_AUX;
template <int R>
__device__ __forceinline__ void rs_store_row(__amdgpu_buffer_rsrc_t rs, const RDiag<R> &C, const RowCtl<R> &ct, int voff) {
    if (lanes_of(ct.mk.lanes)) {
        const int vo = voff + static_cast<int>(ct.soff >> 1);
        if constexpr (R == 1) {
            __builtin_amdgcn_raw_buffer_store_b32(fbits(C.c[0].m), rs, vo, 0, RS_ROW_ST_AUX);
        } else if constexpr (R == 2) {
            __builtin_amdgcn_raw_buffer_store_b64(v2i{fbits(C.c[0].m), fbits(C.c[1].m)}, rs, vo, 0, RS_ROW_ST_AUX);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(C.c[0].m), fbits(C.c[1].m), fbits(C.c[2].m), fbits(C.c[3].m)}, rs, vo, 0, RS_ROW_ST_AUX);
        }
    }
}
template <int R>
struct RFRow {
    float v[R];
};
template <int R>
__device__ __forceinline__ void rs_load_row(__amdgpu_buffer_rsrc_t rs, RFRow<R> &f, const RowCtl<R> &ct, int voff) {
    // Every lane loads, band or not: what a lane outside the band reads (a neighbouring row's bytes, the arena's padding) is never
    // looked at -- rs_emit_pairs masks its hits with the band -- and a load the compiler knows to be issued on every path lets it
    // wait for the OLDER of two rows in flight (s_waitcnt vmcnt(1)) instead of for both: behind a lane-mask branch it had to assume
    // the younger load might not exist and waited for everything, one step after the issue instead of two.
    {
        const int vo = voff + static_cast<int>(ct.soff >> 1);
        if constexpr (R == 1) {
            f.v[0] = bitsf(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, 0, RS_ROW_LD_AUX));
        } else if constexpr (R == 2) {
            const v2i q = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, 0, RS_ROW_LD_AUX);
            f.v[0] = bitsf(q.x), f.v[1] = bitsf(q.y);
        } else {
            const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, RS_ROW_LD_AUX);
            f.v[0] = bitsf(q.x), f.v[1] = bitsf(q.y), f.v[2] = bitsf(q.z), f.v[3] = bitsf(q.w);
        }
    }
}

// posteriors of one anti-diagonal: F * (B * 2^s) / totMant with s = eF + eB - eTot, wave-uniform.  F and B each span fp32's
// whole range, so their product may not be formed first; B * 2^s stays below 2^(RS_TOP + 6 + s), far from overflow while s
// is below NPR_RS_S_LIMIT -- and a task with a row above the limit is run again anyway (npr_device.h).
template <typename T>
__device__ __forceinline__ T &rs_at(T *base, uint32_t byte_off) { return *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off); }
__device__ __forceinline__ float rs_posterior(float f, float b, int s, float inv_tot) { return (f * __builtin_ldexpf(b, s)) * inv_tot; }
template <int R>
__device__ __forceinline__ void rs_emit_pairs(const PairSink &S, const RDiag<R> &B, const RFRow<R> &f, int d, int x0, int y0, const Masks<R> &mk,
                                              int s, float inv_tot, const int (&jr)[R], int &cnt) {
    float p[R];
    uint64_t hit[R], any = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        p[r] = rs_posterior(f.v[r], B.c[r].m, s, inv_tot);
        hit[r] = __ballot(p[r] >= S.threshold) & mk.cell[r];
        any |= hit[r];
    }
    if (d >= 2 && any) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (hit[r]) {
                const int before = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(hit[r] >> 32),
                                                             __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(hit[r]), 0));
                const int slot = cnt + before;
                if (lanes_of(hit[r]) && slot < S.cap) {  // (S.off is 0: the sink's pointers are the task's; unsigned slots: scalar base + 32-bit offset)
                    const uint32_t u = static_cast<uint32_t>(slot) << 2;  // a byte offset that fits 32 bits (pair_cap < 2^29): one shift, the arrays' addresses stay scalar
#if NPR_RS_PAIR_NT
                    __builtin_nontemporal_store(x0 + jr[r] - 1 + S.xs, &rs_at<int32_t>(S.px, u));
                    __builtin_nontemporal_store(y0 - jr[r] - 1 + S.ys, &rs_at<int32_t>(S.py, u));
                    __builtin_nontemporal_store(p[r], &rs_at<float>(S.pp, u));
#else
                    rs_at<int32_t>(S.px, u) = x0 + jr[r] - 1 + S.xs;
                    rs_at<int32_t>(S.py, u) = y0 - jr[r] - 1 + S.ys;
                    rs_at<float>(S.pp, u) = p[r];
#endif
                }
                cnt += __popcll(hit[r]);
            }
        }
    }
}

// ---- what the sweeps of k_dp_rs and k_dp_mid_rs share around the steps ----
// Exponents of the stored rows, one per RS_K anti-diagonals: written by lane 0 with vector stores during a sweep, read back by the
// other sweep through the scalar cache (one s_load per block) after an s_dcache_inv -- the region is reused from task to task, so
// the cache may hold the previous task's words.  (Read with a vector load and handed out by v_readlane, every anti-diagonal waited
// for vmcnt(0): the compiler cannot know that the register is not the target of a load in flight, and the rows prefetched for
// the next step were.)
typedef const __attribute__((address_space(4))) int *cptr_i32;
// The control words of two anti-diagonals, d and d + 1, by ONE scalar load, issued a loop iteration ahead of their use.  (Loaded when
// needed, every step began with s_load + s_waitcnt lgkmcnt(0) -- scalar loads return out of order, so the wait covers the emission
// look-ups in flight too; 64 rows by one vector load handed out by v_readlane was slower still: 3.19 / 3.44 / 3.66e11 cells/s.)
struct CtlPair {
    uint32_t a0, a1, b0, b1;
};
__device__ __forceinline__ CtlPair ctl_scalar2(cptr32 ctl, int d) {
    cptr32 e = ctl + 2 * static_cast<int64_t>(d);
    return CtlPair{e[0], e[1], e[2], e[3]};
}
__device__ __forceinline__ int note_s(int &smax, int s) {
    smax = max(smax, s);
    return s;
}
template <int R>
__device__ __forceinline__ int ctl_rebase_of(uint32_t w1) {  // the rebase a control word asks for, whatever the class's word format
    return R == 2 ? static_cast<int>((w1 >> 28) & 3u) - 1 : static_cast<int>((w1 >> 26) & 3u) - 1;
}
#ifndef NPR_RS_WAVES2
#define NPR_RS_WAVES2 7  // wavefronts per SIMD the R = 2 kernel is compiled for: 72 VGPRs, six spilled outside the sweeps' loops (round 4, without the
                         // short-gap switch terms: 138.1 ms at 7 per SIMD, 140.4 at 6, 138.6 at 8 on the headline batch; round 3, with them: 6 was best)
#endif
#define RS_FWD_REBASE(r) rs_fwd_rebase<R>(E, (r), Q)
#define RS_BWD_REBASE(r) rs_bwd_rebase<R>(E, (r), Q)

}  // namespace

}  // namespace npr
