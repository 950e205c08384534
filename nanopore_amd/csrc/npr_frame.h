// npr_frame.h -- device helpers shared by the register kernels (npr_kernel_stair.hip, npr_kernel_tile.hip):
// uniform-value helpers, DPP neighbour moves, the anti-diagonal held in registers (Diag<R>), base streams, band lane
// masks built on the scalar unit, forward rows through raw buffer descriptors, posterior emission, and the X / Y steps
// of the frame-based sweep.  Everything lives in an anonymous namespace of the including translation unit.
#pragma once
#include <hip/hip_runtime.h>

#include "npr_cell.h"
#include "npr_device.h"

namespace npr {

namespace {

constexpr int WAVE = 64;
#ifndef NPR_T_SGPR_MIN_R
#define NPR_T_SGPR_MIN_R 2  // transitions in SGPRs from this many slots per lane on (below: VGPRs)
#endif
constexpr int MODEL_FLOATS = sizeof(DevModel) / sizeof(float);

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float unif(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int64_t uni64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32));
    return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}
__device__ __forceinline__ int fbits(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ float bitsf(int v) { return __builtin_bit_cast(float, v); }

// lane l <- lane l+1 (lane 63 takes `edge`);  lane l <- lane l-1 (lane 0 takes `edge`)
__device__ __forceinline__ int dpp_from_above(int v, int edge) {
    return __builtin_amdgcn_update_dpp(edge, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_from_below(int v, int edge) {
    return __builtin_amdgcn_update_dpp(edge, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// mantissas: the edge lane takes 0 (bound_ctrl), which needs no `old` register
__device__ __forceinline__ float dppf_from_above(float v) {
    return bitsf(__builtin_amdgcn_update_dpp(0, fbits(v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float dppf_from_below(float v) {
    return bitsf(__builtin_amdgcn_update_dpp(0, fbits(v), 0x138, 0xf, 0xf, true));
}

template <int R>
struct Diag {  // one anti-diagonal in registers: slot j = R*lane + r
    Cell c[R];
};

template <int R>
__device__ __forceinline__ Diag<R> dead_diag() {
    Diag<R> d;
#pragma unroll
    for (int r = 0; r < R; ++r) d.c[r] = dead_cell();
    return d;
}

// out[j] = in[j+1]
template <int R>
__device__ __forceinline__ Diag<R> shift_up(const Diag<R> &in) {
    Diag<R> o;
#pragma unroll
    for (int r = 0; r + 1 < R; ++r) o.c[r] = in.c[r + 1];
    o.c[R - 1].m = dppf_from_above(in.c[0].m);
    o.c[R - 1].sx = dppf_from_above(in.c[0].sx);
    o.c[R - 1].sy = dppf_from_above(in.c[0].sy);
    o.c[R - 1].lx = dppf_from_above(in.c[0].lx);
    o.c[R - 1].ly = dppf_from_above(in.c[0].ly);
    o.c[R - 1].e = dpp_from_above(in.c[0].e, E_DEAD);
    return o;
}
// out[j] = in[j-1]
template <int R>
__device__ __forceinline__ Diag<R> shift_down(const Diag<R> &in) {
    Diag<R> o;
#pragma unroll
    for (int r = 1; r < R; ++r) o.c[r] = in.c[r - 1];
    o.c[0].m = dppf_from_below(in.c[R - 1].m);
    o.c[0].sx = dppf_from_below(in.c[R - 1].sx);
    o.c[0].sy = dppf_from_below(in.c[R - 1].sy);
    o.c[0].lx = dppf_from_below(in.c[R - 1].lx);
    o.c[0].ly = dppf_from_below(in.c[R - 1].ly);
    o.c[0].e = dpp_from_below(in.c[R - 1].e, E_DEAD);
    return o;
}

// base codes pre-multiplied by 4 (byte offsets into the LDS tables); code 4 (N) = 16
template <int R>
struct Bases {
    int b[R];
};
// b[j] <- b[j+1], the top slot takes `inject`
template <int R>
__device__ __forceinline__ void bases_up(Bases<R> &s, int inject) {
    const int first = s.b[0];
#pragma unroll
    for (int r = 0; r + 1 < R; ++r) s.b[r] = s.b[r + 1];
    s.b[R - 1] = dpp_from_above(first, inject);
}
// b[j] <- b[j-1], slot 0 takes `inject`
template <int R>
__device__ __forceinline__ void bases_down(Bases<R> &s, int inject) {
    const int last = s.b[R - 1];
#pragma unroll
    for (int r = R - 1; r > 0; --r) s.b[r] = s.b[r - 1];
    s.b[0] = dpp_from_below(last, inject);
}

__device__ __forceinline__ int base4(const uint8_t *seq, int len, int idx) {
    return (idx >= 0 && idx < len) ? 4 * static_cast<int>(seq[idx]) : 16;
}

// A 64-base block of a sequence held one base per lane, with the next block prefetched.
// dir = +1: lane l holds seq[base + l]; dir = -1: lane l holds seq[base - l].
struct Feed {
    int cur, nxt;  // per-lane base*4
    int base;      // uniform: index held by lane 0 of `cur`
};
template <int DIR>
__device__ __forceinline__ void feed_init(Feed &f, const uint8_t *seq, int len, int first, int lane) {
    f.base = first;
    f.cur = base4(seq, len, first + DIR * lane);
    f.nxt = base4(seq, len, first + DIR * (64 + lane));
}
// base*4 of sequence index `idx` (uniform), which must move monotonically in direction DIR
template <int DIR>
__device__ __forceinline__ int feed_get(Feed &f, const uint8_t *seq, int len, int idx, int lane) {
    // (uni: in k_dp_wide the feed lives in a struct the step lambdas capture, and the compiler no longer sees that
    // `base` is wave-uniform -- without it the refill test becomes per-lane code with an exec mask)
    int off = uni(DIR * (idx - f.base));
    if (off >= 64) {  // uniform
        f.cur = f.nxt;
        f.base += DIR * 64;
        f.nxt = base4(seq, len, f.base + DIR * (64 + lane));
        off -= 64;
    }
    return __builtin_amdgcn_readlane(f.cur, off);
}

// The base a feed would deliver for `idx` (at most one step ahead of its last request), without moving the feed: a
// later request may be for idx + 1 again (a rebase in between), which a block switch made here would have lost.
template <int DIR>
__device__ __forceinline__ int feed_peek(const Feed &f, int idx) {
    const int off = uni(DIR * (idx - f.base));
    return off < 64 ? __builtin_amdgcn_readlane(f.cur, off) : __builtin_amdgcn_readlane(f.nxt, off - 64);
}

typedef const __attribute__((address_space(4))) uint32_t *cptr32;

// Control word of one anti-diagonal (stair_step, npr_sched.h; made by k_plan_sched when a batch is staged): where the band sits in the frame,
// which step brought the frame here, and where the row starts in the forward scratch.
struct Ctl {
    uint32_t co;  // scratch offset (cells) of the first stored lane of the row
    int jlo, n;   // band = slots [jlo, jlo + n)
    int reb;      // frame rebase applied between the previous anti-diagonal and the step into this one: -1, 0, +1
};
__device__ __forceinline__ Ctl read_ctl(cptr32 ctl, int d) {
    const uint32_t co = ctl[2 * d], w = ctl[2 * d + 1];
    return Ctl{co, static_cast<int>(w & 8191u), static_cast<int>((w >> 13) & 8191u), static_cast<int>((w >> 26) & 3u) - 1};
}

__device__ __forceinline__ uint64_t low_lanes(int k) { return k >= 64 ? ~0ull : ((1ull << k) - 1ull); }

// Wave-uniform lane masks of a band inside the frame (scalar unit only).
template <int R>
struct Masks {
    uint64_t cell[R];  // lanes whose slot R*lane + r is inside the band
    uint64_t lanes;    // lanes holding at least one band cell
    int l0;            // first such lane
};
template <int R>
__device__ __forceinline__ Masks<R> band_masks(int jlo, int n) {
    constexpr int SH = R == 1 ? 0 : (R == 2 ? 1 : 2);
    Masks<R> m;
#pragma unroll
    for (int r = 0; r < R; ++r) m.cell[r] = low_lanes((jlo + n - r + R - 1) >> SH) & ~low_lanes((jlo - r + R - 1) >> SH);
    m.l0 = jlo >> SH;
    m.lanes = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) m.lanes |= m.cell[r];
    return m;
}

// What k_dp_stair needs to know about one anti-diagonal, ready to use: the lane masks of the band, the byte offset of
// the row in the task's forward scratch (where lane 0 would land, biased by row_bias so that it is never negative: the
// byte offset added to every lane's own) and the rebase that leads into it.  For R = 2 -- the north-star class
// -- the control words hold exactly that (npr_sched.h, stair_packed): four bit-field extracts and two s_bfm_b64 instead
// of the ~40 scalar instructions that (jlo, n, co) took.  The other classes build the same from their control words.
template <int R>
struct RowCtl {
    Masks<R> mk;
    uint32_t soff;
    int reb;
    int jlo;
    uint32_t moved;  // non-zero: the band is not where it was two anti-diagonals ago (bit 30 of the control word, npr_sched.h)
};
// lanes [lo, lo + w): s_bfm_b64 takes the low six bits of both operands (w <= 63; lo = 64 comes with w = 0), so the
// bit fields of the control word need no masking -- one shift per field and this
__device__ __forceinline__ uint64_t lane_run(uint32_t lo, uint32_t w) {
    uint64_t m;
    asm("s_bfm_b64 %0, %1, %2" : "=s"(m) : "s"(w), "s"(lo));
    return m;
}
template <int R>
__device__ __forceinline__ RowCtl<R> row_ctl_of_words(uint32_t w0, uint32_t w1);
template <int R>
__device__ __forceinline__ RowCtl<R> read_row_ctl_at(cptr32 e) {  // e: the anti-diagonal's two control words
    RowCtl<R> c;
    if constexpr (R == 2) {
        const uint32_t so = e[0], w = e[1];
        c.mk.cell[0] = lane_run(w, w >> 14);
        c.mk.cell[1] = lane_run(w >> 7, w >> 21);
        c.mk.lanes = c.mk.cell[0] | c.mk.cell[1];
        c.mk.l0 = static_cast<int>((w >> 7) & 127u);
        c.soff = so;
        c.reb = static_cast<int>((w >> 28) & 3u) - 1;
        c.jlo = static_cast<int>((w & 127u) + ((w >> 7) & 127u));
        c.moved = w & (1u << 30);
    } else {
        const Ctl t = read_ctl(e, 0);
        c.mk = band_masks<R>(t.jlo, t.n);
        c.soff = ((t.co - static_cast<uint32_t>(R * c.mk.l0)) << 3) + row_bias<R>();
        c.reb = t.reb;
        c.jlo = t.jlo;
        c.moved = e[1] & (1u << 30);
    }
    return c;
}
template <int R>
__device__ __forceinline__ RowCtl<R> read_row_ctl(cptr32 ctl, int d) { return read_row_ctl_at<R>(ctl + 2 * static_cast<int64_t>(d)); }
// the same from the two words themselves (wave-uniform values that did not come through the scalar cache: k_dp_pair's
// posterior pass fetches the words of 64 anti-diagonals with one vector load and hands them out by v_readlane)
template <int R>
__device__ __forceinline__ RowCtl<R> row_ctl_of_words(uint32_t w0, uint32_t w1) {
    RowCtl<R> c;
    if constexpr (R == 2) {
        c.mk.cell[0] = lane_run(w1, w1 >> 14);
        c.mk.cell[1] = lane_run(w1 >> 7, w1 >> 21);
        c.mk.lanes = c.mk.cell[0] | c.mk.cell[1];
        c.mk.l0 = static_cast<int>((w1 >> 7) & 127u);
        c.soff = w0;
        c.reb = static_cast<int>((w1 >> 28) & 3u) - 1;
        c.jlo = static_cast<int>((w1 & 127u) + ((w1 >> 7) & 127u));
        c.moved = w1 & (1u << 30);
    } else {
        const Ctl t{w0, static_cast<int>(w1 & 8191u), static_cast<int>((w1 >> 13) & 8191u), static_cast<int>((w1 >> 26) & 3u) - 1};
        c.mk = band_masks<R>(t.jlo, t.n);
        c.soff = ((t.co - static_cast<uint32_t>(R * c.mk.l0)) << 3) + row_bias<R>();
        c.reb = t.reb;
        c.jlo = t.jlo;
        c.moved = w1 & (1u << 30);
    }
    return c;
}
// a packed control word (one-wavefront R = 2 tasks) in the terms of the other kernels (k_em_stair<2>)
__device__ __forceinline__ Ctl read_ctl_packed(cptr32 ctl, int d) {
    const uint32_t so = ctl[2 * d], w = ctl[2 * d + 1];
    const uint32_t lo0 = w & 127u, lo1 = (w >> 7) & 127u;
    return Ctl{(((so - row_bias<2>()) >> 3) + 2u * lo1) & ((1u << 29) - 1u), static_cast<int>(lo0 + lo1), static_cast<int>(((w >> 14) & 127u) + ((w >> 21) & 127u)),
               static_cast<int>((w >> 28) & 3u) - 1};
}

// control word of a one-wavefront task of class R
template <int R>
__device__ __forceinline__ Ctl read_ctl_one(cptr32 ctl, int d) {
    if constexpr (R == 2) return read_ctl_packed(ctl, d);
    else return read_ctl(ctl, d);
}

// Everything wave-uniform a step needs.
struct StepEnv {
    const DevModel *mdl;
    const char *ltab;
    Trans tr;
    const uint8_t *X, *Y;
    int lX, lY;
    int lane;
};

template <int R>
__device__ __forceinline__ void emissions(const StepEnv &E, const Bases<R> &bx, const Bases<R> &by, int r, float &em,
                                          float &exs, float &exl, float &eys, float &eyl) {
    constexpr int OFF_EM = offsetof(DevModel, em), OFF_EX = offsetof(DevModel, ex), OFF_EY = offsetof(DevModel, ey);
#ifdef NPR_EMIDX_MAD24
    em = *reinterpret_cast<const float *>(E.ltab + OFF_EM + (__umul24(static_cast<unsigned>(bx.b[r]), 5u) + static_cast<unsigned>(by.b[r])));
#else
    em = *reinterpret_cast<const float *>(E.ltab + OFF_EM + 5 * bx.b[r] + by.b[r]);
#endif
    exs = *reinterpret_cast<const float *>(E.ltab + OFF_EX + 20 + bx.b[r]);
    exl = *reinterpret_cast<const float *>(E.ltab + OFF_EX + 60 + bx.b[r]);
    eys = *reinterpret_cast<const float *>(E.ltab + OFF_EY + 40 + by.b[r]);
    eyl = *reinterpret_cast<const float *>(E.ltab + OFF_EY + 80 + by.b[r]);
}

// A slot outside the band keeps whatever mantissas the arithmetic produced and only gets the dead exponent: every
// consumer multiplies it by scale2(E_DEAD - eref) = 0, so the mantissas never matter.
__device__ __forceinline__ bool lanes_of(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
__device__ __forceinline__ void kill_outside(Cell &c, uint64_t in_band) {
    c.e = __builtin_amdgcn_inverse_ballot_w64(in_band) ? c.e : E_DEAD;
}

// The end of a step: renormalise the new cells if this anti-diagonal, d, does (wave-uniform: norm_diag(d),
// npr_cell.h -- one branch on the scalar unit around 12 instructions per cell) and mark the slots outside the band dead.
template <int R>
__device__ __forceinline__ void settle_diag(int d, Diag<R> &o, const Masks<R> &mk) {
    // (one `if`, no `else`: with both arms the compiler lays the arms out one after the other behind a 64-bit flag and
    // spends six scalar instructions and two branches on it)
    if (norm_diag(d)) {
#pragma unroll
        for (int r = 0; r < R; ++r) normalise(o.c[r], o.c[r].e);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) kill_outside(o.c[r], mk.cell[r]);
}

// In-place moves of the whole register state by one slot (frame rebase).  Written as inline assembly on tied
// operands: expressed in C++ the moved values are new SSA values, and the compiler pays for the join with the
// not-moved path by copying the state on the hot path.  s_nop: a DPP read of a VGPR written by the previous VALU
// instruction needs two wait states, which the compiler cannot see through inline assembly.
__device__ __forceinline__ void rot_up(float &a, float &b) { asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void rot_up(int &a, int &b) { asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void dpp_up_inplace(float &v) {
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v));
}
__device__ __forceinline__ void dpp_down_inplace(float &v) {
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v));
}
// integer registers: the vacated edge lane takes `edge` (uniform)
__device__ __forceinline__ void dpp_up_inplace(int &v, int edge_) {
    const int edge = uni(edge_);
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_writelane_b32 %0, %1, 63" : "+v"(v) : "s"(edge));
}
__device__ __forceinline__ void dpp_down_inplace(int &v, int edge_) {
    const int edge = uni(edge_);
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_writelane_b32 %0, %1, 0" : "+v"(v) : "s"(edge));
}

// slot j <- slot j+1
template <int R>
__device__ __forceinline__ void diag_up_inplace(Diag<R> &g) {
#pragma unroll
    for (int r = 0; r + 1 < R; ++r) {  // rotate: (c0, c1, .., c{R-1}) -> (c1, .., c{R-1}, c0)
        rot_up(g.c[r].m, g.c[r + 1].m), rot_up(g.c[r].sx, g.c[r + 1].sx), rot_up(g.c[r].sy, g.c[r + 1].sy);
        rot_up(g.c[r].lx, g.c[r + 1].lx), rot_up(g.c[r].ly, g.c[r + 1].ly), rot_up(g.c[r].e, g.c[r + 1].e);
    }
    Cell &t = g.c[R - 1];
    dpp_up_inplace(t.m), dpp_up_inplace(t.sx), dpp_up_inplace(t.sy), dpp_up_inplace(t.lx), dpp_up_inplace(t.ly);
    dpp_up_inplace(t.e, E_DEAD);
}
// slot j <- slot j-1
template <int R>
__device__ __forceinline__ void diag_down_inplace(Diag<R> &g) {
#pragma unroll
    for (int r = R - 1; r > 0; --r) {  // rotate: (c0, .., c{R-1}) -> (c{R-1}, c0, .., c{R-2})
        rot_up(g.c[r].m, g.c[r - 1].m), rot_up(g.c[r].sx, g.c[r - 1].sx), rot_up(g.c[r].sy, g.c[r - 1].sy);
        rot_up(g.c[r].lx, g.c[r - 1].lx), rot_up(g.c[r].ly, g.c[r - 1].ly), rot_up(g.c[r].e, g.c[r - 1].e);
    }
    Cell &t = g.c[0];
    dpp_down_inplace(t.m), dpp_down_inplace(t.sx), dpp_down_inplace(t.sy), dpp_down_inplace(t.lx), dpp_down_inplace(t.ly);
    dpp_down_inplace(t.e, E_DEAD);
}
template <int R>
__device__ __forceinline__ void bases_up_inplace(Bases<R> &s, int inject) {
#pragma unroll
    for (int r = 0; r + 1 < R; ++r) rot_up(s.b[r], s.b[r + 1]);
    dpp_up_inplace(s.b[R - 1], inject);
}
template <int R>
__device__ __forceinline__ void bases_down_inplace(Bases<R> &s, int inject) {
#pragma unroll
    for (int r = R - 1; r > 0; --r) rot_up(s.b[r], s.b[r - 1]);
    dpp_down_inplace(s.b[0], inject);
}

// Base streams of one sweep plus what a rebase needs to run them backwards by one slot: the base that left the
// wavefront at the last step of each kind (a rebase towards higher x-y only ever follows a Y-step, one towards lower
// x-y an X-step: stair_step, npr_sched.h).
template <int R>
struct Streams {
    Bases<R> X, Y;
    Feed fx, fy;
    int xcap, ycap;
};

// Frame rebase of the forward sweep, r = +1: (x0, y0) -> (x0 + 1, y0 - 1), every slot takes its upper neighbour.
template <int R>
__device__ __forceinline__ void fwd_rebase(const StepEnv &E, int r, Diag<R> &A, Diag<R> &B, Streams<R> &S, int &x0, int &y0) {
    if (r > 0) {
        diag_up_inplace<R>(A), diag_up_inplace<R>(B);
        x0 += 1, y0 -= 1;
        bases_up_inplace<R>(S.X, feed_get<+1>(S.fx, E.X, E.lX, x0 + 64 * R - 2, E.lane));
        bases_up_inplace<R>(S.Y, S.ycap);
    } else {
        diag_down_inplace<R>(A), diag_down_inplace<R>(B);
        x0 -= 1, y0 += 1;
        bases_down_inplace<R>(S.X, S.xcap);
        bases_down_inplace<R>(S.Y, feed_get<+1>(S.fy, E.Y, E.lY, y0 - 1, E.lane));
    }
}
// ... and of the backward sweep, which undoes the forward one: r is the forward rebase being undone.
template <int R>
__device__ __forceinline__ void bwd_rebase(const StepEnv &E, int r, Diag<R> &A, Diag<R> &B, Streams<R> &S, int &x0, int &y0) {
    if (r > 0) {  // back to lower x-y: (x0 - 1, y0 + 1)
        diag_down_inplace<R>(A), diag_down_inplace<R>(B);
        x0 -= 1, y0 += 1;
        bases_down_inplace<R>(S.X, feed_get<-1>(S.fx, E.X, E.lX, x0, E.lane));
        bases_down_inplace<R>(S.Y, S.ycap);
    } else {
        diag_up_inplace<R>(A), diag_up_inplace<R>(B);
        x0 += 1, y0 -= 1;
        bases_up_inplace<R>(S.X, S.xcap);
        bases_up_inplace<R>(S.Y, feed_get<-1>(S.fy, E.Y, E.lY, y0 - (64 * R - 1), E.lane));
    }
}

// One forward anti-diagonal.  `io` holds anti-diagonal d-2 on entry and d on exit; `p1` holds d-1.  S.X / S.Y hold
// X[x-1]*4 and Y[y-1]*4 of every slot.  d: the anti-diagonal being computed (norm_diag(d) says whether it renormalises).
template <int R>
__device__ __forceinline__ void fwd_x_step(int d, const StepEnv &E, Diag<R> &io, const Diag<R> &p1, Streams<R> &S, int &x0, const Masks<R> &mk) {
    S.xcap = __builtin_amdgcn_readlane(S.X.b[0], 0);
    x0 += 1;
    bases_up<R>(S.X, feed_get<+1>(S.fx, E.X, E.lX, x0 + 64 * R - 2, E.lane));
    const Diag<R> U = shift_up<R>(p1);  // (x, y-1) is slot j+1 of d-1; (x-1, y) keeps slot j
    Diag<R> o;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float em, exs, exl, eys, eyl;
        emissions<R>(E, S.X, S.Y, r, em, exs, exl, eys, eyl);
        o.c[r] = fwd_cell<false>(E.tr, p1.c[r], io.c[r], U.c[r], em, exs, exl, eys, eyl);
    }
    settle_diag<R>(d, o, mk);
    io = o;
}
template <int R>
__device__ __forceinline__ void fwd_y_step(int d, const StepEnv &E, Diag<R> &io, const Diag<R> &p1, Streams<R> &S, int &y0, const Masks<R> &mk) {
    S.ycap = __builtin_amdgcn_readlane(S.Y.b[R - 1], 63);
    y0 += 1;
    bases_down<R>(S.Y, feed_get<+1>(S.fy, E.Y, E.lY, y0 - 1, E.lane));
    const Diag<R> L = shift_down<R>(p1);  // (x-1, y) is slot j-1 of d-1; (x, y-1) keeps slot j
    Diag<R> o;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float em, exs, exl, eys, eyl;
        emissions<R>(E, S.X, S.Y, r, em, exs, exl, eys, eyl);
        o.c[r] = fwd_cell<false>(E.tr, L.c[r], io.c[r], p1.c[r], em, exs, exl, eys, eyl);
    }
    settle_diag<R>(d, o, mk);
    io = o;
}

// Forward rows in HBM: a row holds the lanes [l0, l1) that carry band cells, 8R bytes per lane -- per slot the pair
// (match mantissa, exponent).  A row is addressed through a raw buffer descriptor rebuilt per row on the scalar unit
// (base = where lane 0 would land) with a per-lane constant offset, under the row's lane mask: one vector-memory
// instruction per row (two for R = 4), no per-lane address arithmetic.
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int R>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(char *F, uint32_t co, int l0) {
    return __builtin_amdgcn_make_buffer_rsrc(F + (static_cast<int64_t>(co) - R * l0) * 8, 0, -1, 0x00020000);
}

template <int R>
__device__ __forceinline__ void store_row(char *F, const Diag<R> &C, const Ctl &ct, int voff) {
    const Masks<R> mk = band_masks<R>(ct.jlo, ct.n);
    const __amdgpu_buffer_rsrc_t rs = row_rsrc<R>(F, ct.co, mk.l0);
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

// ... and with the descriptor of the task's scratch made once (base = F - row_bias) and the row's offset ready-made in
// the control word: one vector add per row instead of seven scalar instructions.  (The offset goes into the VECTOR
// offset, not the instruction's scalar offset: with the row offset in a short-lived SGPR as soffset, the next scalar
// instruction reused that register and about one launch in a hundred read rows from a wrong address when another kernel
// kept the memory pipeline busy -- tools/stress_tile.py; operands in VGPRs are read before the wavefront moves on.)
template <int R>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t task_rsrc(char *F) {
    return __builtin_amdgcn_make_buffer_rsrc(F - static_cast<int64_t>(row_bias<R>()), 0, -1, 0x00020000);
}
template <int R>
__device__ __forceinline__ void store_row(__amdgpu_buffer_rsrc_t rs, const Diag<R> &C, const RowCtl<R> &ct, int voff) {
    if (__builtin_amdgcn_inverse_ballot_w64(ct.mk.lanes)) {
        const int vo = voff + static_cast<int>(ct.soff);
        if constexpr (R == 1) {
            __builtin_amdgcn_raw_buffer_store_b64(v2i{fbits(C.c[0].m), C.c[0].e}, rs, vo, 0, 0);
        } else if constexpr (R == 2) {
            __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(C.c[0].m), C.c[0].e, fbits(C.c[1].m), C.c[1].e}, rs, vo, 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(C.c[0].m), C.c[0].e, fbits(C.c[1].m), C.c[1].e}, rs, vo, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(v4i{fbits(C.c[2].m), C.c[2].e, fbits(C.c[3].m), C.c[3].e}, rs, vo + 16, 0, 0);
        }
    }
}

template <int R>
struct FRow {  // forward match values of one anti-diagonal
    float v[R];
    int e[R];
};

// lanes outside the row keep stale registers: every consumer masks by the band
template <int R>
__device__ __forceinline__ void load_row(char *F, FRow<R> &f, const Ctl &ct, int voff) {
    const Masks<R> mk = band_masks<R>(ct.jlo, ct.n);
    const __amdgpu_buffer_rsrc_t rs = row_rsrc<R>(F, ct.co, mk.l0);
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

template <int R>
__device__ __forceinline__ void load_row(__amdgpu_buffer_rsrc_t rs, FRow<R> &f, const RowCtl<R> &ct, int voff) {
    if (__builtin_amdgcn_inverse_ballot_w64(ct.mk.lanes)) {
        const int vo = voff + static_cast<int>(ct.soff);
        if constexpr (R == 1) {
            const v2i q = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, 0, 0);
            f.v[0] = bitsf(q.x), f.e[0] = q.y;
        } else if constexpr (R == 2) {
            const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0);
            f.v[0] = bitsf(q.x), f.e[0] = q.y, f.v[1] = bitsf(q.z), f.e[1] = q.w;
        } else {
            const v4i q = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0);
            const v4i g = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 16, 0, 0);
            f.v[0] = bitsf(q.x), f.e[0] = q.y, f.v[1] = bitsf(q.z), f.e[1] = q.w;
            f.v[2] = bitsf(g.x), f.e[2] = g.y, f.v[3] = bitsf(g.z), f.e[3] = g.w;
        }
    }
}

struct PairSink {
    int32_t *px, *py;
    float *pp;
    int64_t off;
    int cap, xs, ys;
    float threshold;
};

// posteriors of one anti-diagonal (d >= 2: from there on the forward match value is zero wherever x < 1 or y < 1;
// d = 0 is the start cell, whose match state holds the start probability)
template <int R>
__device__ __forceinline__ void emit_pairs(const PairSink &S, const Diag<R> &B, const FRow<R> &f, int d, int x0, int y0,
                                           const Masks<R> &mk, int tot_e, float inv_tot, const int (&jr)[R], int &cnt) {
    float p[R];
    uint64_t hit[R], any = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        p[r] = posterior(f.v[r], f.e[r], B.c[r].m, B.c[r].e, tot_e, inv_tot);
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
                if (__builtin_amdgcn_inverse_ballot_w64(hit[r]) && slot < S.cap) {
                    S.px[S.off + slot] = x0 + jr[r] - 1 + S.xs;
                    S.py[S.off + slot] = y0 - jr[r] - 1 + S.ys;
                    S.pp[S.off + slot] = p[r];
                }
                cnt += __popcll(hit[r]);
            }
        }
    }
}

// One backward anti-diagonal d.  `io` holds anti-diagonal d+2 on entry and d on exit; `s1` holds d+1.  S.X / S.Y hold
// X[x]*4 and Y[y]*4 of every slot.  The X variant undoes the X-step into d+1 (d even), the Y variant a Y-step.
template <int R>
__device__ __forceinline__ void bwd_x_step(int d, const StepEnv &E, Diag<R> &io, const Diag<R> &s1, Streams<R> &S, int &x0, const Masks<R> &mk) {
    S.xcap = __builtin_amdgcn_readlane(S.X.b[R - 1], 63);
    x0 -= 1;
    // x decreased by one in every slot: X[x] moves up a slot, slot 0 takes X[x0]
    bases_down<R>(S.X, feed_get<-1>(S.fx, E.X, E.lX, x0, E.lane));
    const Diag<R> Ys = shift_down<R>(s1);  // (x, y+1) is slot j-1 of d+1; (x+1, y) keeps slot j
    Diag<R> o;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float em, exs, exl, eys, eyl;
        emissions<R>(E, S.X, S.Y, r, em, exs, exl, eys, eyl);
        o.c[r] = bwd_cell<false>(E.tr, io.c[r], s1.c[r], Ys.c[r], em, exs, exl, eys, eyl);
    }
    settle_diag<R>(d, o, mk);
    io = o;
}
template <int R>
__device__ __forceinline__ void bwd_y_step(int d, const StepEnv &E, Diag<R> &io, const Diag<R> &s1, Streams<R> &S, int &y0, const Masks<R> &mk) {
    S.ycap = __builtin_amdgcn_readlane(S.Y.b[0], 0);
    y0 -= 1;
    bases_up<R>(S.Y, feed_get<-1>(S.fy, E.Y, E.lY, y0 - (64 * R - 1), E.lane));
    const Diag<R> Xs = shift_up<R>(s1);  // (x+1, y) is slot j+1 of d+1; (x, y+1) keeps slot j
    Diag<R> o;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float em, exs, exl, eys, eyl;
        emissions<R>(E, S.X, S.Y, r, em, exs, exl, eys, eyl);
        o.c[r] = bwd_cell<false>(E.tr, io.c[r], Xs.c[r], s1.c[r], em, exs, exl, eys, eyl);
    }
    settle_diag<R>(d, o, mk);
    io = o;
}

// the same with the masks built from a control word (k_em_stair)
#define NPR_CTL_STEP(name)                                                                                                     \
    template <int R>                                                                                                           \
    __device__ __forceinline__ void name(int d, const StepEnv &E, Diag<R> &io, const Diag<R> &o, Streams<R> &S, int &c0, \
                                         const Ctl &ct) {                                                                      \
        name<R>(d, E, io, o, S, c0, band_masks<R>(ct.jlo, ct.n));                                                           \
    }
NPR_CTL_STEP(fwd_x_step)
NPR_CTL_STEP(fwd_y_step)
NPR_CTL_STEP(bwd_x_step)
NPR_CTL_STEP(bwd_y_step)
#undef NPR_CTL_STEP

}  // namespace

}  // namespace npr
