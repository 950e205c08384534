// npr_cell.h -- device arithmetic of one DP cell (DESIGN.md "Device arithmetic").
//
// A cell holds the five state probabilities in block floating point: five linear fp32 mantissas v[s]
// sharing one int32 binary exponent e, value_s = v[s] * 2^e, max_s v[s] in [0.5, 1).  That is the
// log-sum-exp recurrence of cactus_realign's five-state machine with the
// integer part of log2 carried exactly and the fractional part carried linearly, so the inner loop needs
// no exp/log and never leaves fp32 range however long the read is.  Every operation is a single-rounding
// IEEE op; oracle/realign_oracle_f32.c restates the same sequence on the CPU and the parity tests demand
// bit-identical results -- keep the two in step.
#pragma once
#include <hip/hip_runtime.h>

#include "npr_device.h"

namespace npr {

struct Cell {
    float m, sx, sy, lx, ly;  // states 0, 1, 2, 3, 4
    int e;
};

__device__ __forceinline__ Cell dead_cell() { return Cell{0.f, 0.f, 0.f, 0.f, 0.f, E_DEAD}; }

// 2^k for -126 <= k <= 0 built from its bit pattern, 0 below (k is an exponent difference, never positive).
// Integer ops only: v_ldexp_f32 issues at about half the rate of a plain VALU op on gfx950 (tools/valu_rates).
__device__ __forceinline__ float scale2(int k) {
    const int t = k + 127;
    return __builtin_bit_cast(float, (t > 0 ? t : 0) << 23);
}

// Renormalise so that the largest mantissa lands in [0.5, 1): multiply by 2^(126 - E) where E is the biased
// exponent field of the largest value (exact: a power of two), and add E - 126 to the shared exponent.
// A zero cell becomes the dead cell.  (A subnormal maximum, E = 0, is scaled by 2^126 and stays below 0.5:
// harmless, and the CPU mirror does exactly the same.)
__device__ __forceinline__ void normalise(Cell &c, int eref) {
    const float vmax = fmaxf(fmaxf(c.m, c.sx), fmaxf(fmaxf(c.sy, c.lx), c.ly));
    const int bits = __builtin_bit_cast(int, vmax) & 0x7f800000;
    const float inv = __builtin_bit_cast(float, 0x7e800000 - bits);  // 2^(126 - E)
    c.m *= inv;
    c.sx *= inv;
    c.sy *= inv;
    c.lx *= inv;
    c.ly *= inv;
    c.e = vmax > 0.0f ? eref + (bits >> 23) - 126 : E_DEAD;
}

// Which anti-diagonals renormalise: d = 0, 1 (mod 4).  On d = 2, 3 (mod 4) a cell keeps the reference exponent of
// its predecessors and whatever mantissas the recurrence produced.  Every predecessor chain (steps of 1 or 2
// anti-diagonals, either sweep direction) meets a renormalising anti-diagonal after at most two such cells, so the
// mantissas stay far inside fp32's range (each step costs at most one transition, one emission and one
// exponent spread), and half of the renormalisations -- 12 of the ~55 VALU instructions of a cell -- are not issued.
// (Skipping every second anti-diagonal instead would not do: match moves step by two, so the even anti-diagonals
// would never renormalise along a run of matches.)  The CPU mirror follows the same rule.
#ifndef NPR_NORM_MASK
#define NPR_NORM_MASK 2
#endif
__host__ __device__ constexpr bool norm_diag(int d) { return (d & NPR_NORM_MASK) == 0; }

template <bool NORM>
__device__ __forceinline__ void settle(Cell &c, int eref) {
    if constexpr (NORM) normalise(c, eref);
    else c.e = eref;
}

// transition probabilities held in registers (wave-uniform)
struct Trans {
    float mm, sxm, sym, lxm, lym;  // -> match
    float msx, sxsx, sysx;         // -> shortGapX
    float msy, sysy, sxsy;         // -> shortGapY
    float mlx, lxlx;               // -> longGapX
    float mly, lyly;               // -> longGapY
};

__device__ __forceinline__ Trans load_trans(const float *T) {
    Trans t;
    t.mm = T[0], t.msx = T[1], t.msy = T[2], t.mlx = T[3], t.mly = T[4];
    t.sxm = T[5], t.sxsx = T[6], t.sxsy = T[7];
    t.sym = T[10], t.sysx = T[11], t.sysy = T[12];
    t.lxm = T[15], t.lxlx = T[18];
    t.lym = T[20], t.lyly = T[24];
    return t;
}

// forward: L = (x-1,y), M = (x-1,y-1), U = (x,y-1); em/exs/exl/eys/eyl the emissions of the bases consumed
template <bool NORM = true>
__device__ __forceinline__ Cell fwd_cell(const Trans &t, const Cell &L, const Cell &M, const Cell &U, float em,
                                         float exs, float exl, float eys, float eyl) {
    const int eref = max(L.e, max(M.e, U.e));
    const float fL = scale2(L.e - eref), fM = scale2(M.e - eref), fU = scale2(U.e - eref);
    Cell c;
    float a;
    a = t.mm * M.m;
    a = __builtin_fmaf(t.sxm, M.sx, a);
    a = __builtin_fmaf(t.sym, M.sy, a);
    a = __builtin_fmaf(t.lxm, M.lx, a);
    a = __builtin_fmaf(t.lym, M.ly, a);
    c.m = (fM * em) * a;
    a = t.msx * L.m;
    a = __builtin_fmaf(t.sxsx, L.sx, a);
#ifndef NPR_NO_SHORT_SWITCH
    a = __builtin_fmaf(t.sysx, L.sy, a);
#endif
    c.sx = (fL * exs) * a;
    a = t.mlx * L.m;
    a = __builtin_fmaf(t.lxlx, L.lx, a);
    c.lx = (fL * exl) * a;
    a = t.msy * U.m;
    a = __builtin_fmaf(t.sysy, U.sy, a);
#ifndef NPR_NO_SHORT_SWITCH
    a = __builtin_fmaf(t.sxsy, U.sx, a);
#endif
    c.sy = (fU * eys) * a;
    a = t.mly * U.m;
    a = __builtin_fmaf(t.lyly, U.ly, a);
    c.ly = (fU * eyl) * a;
    settle<NORM>(c, eref);
    return c;
}

// backward: Ms = (x+1,y+1), Xs = (x+1,y), Ys = (x,y+1); emissions of the bases those moves consume
template <bool NORM = true>
__device__ __forceinline__ Cell bwd_cell(const Trans &t, const Cell &Ms, const Cell &Xs, const Cell &Ys, float em,
                                         float exs, float exl, float eys, float eyl) {
    const int eref = max(Ms.e, max(Xs.e, Ys.e));
    const float fM = scale2(Ms.e - eref), fX = scale2(Xs.e - eref), fY = scale2(Ys.e - eref);
    const float am = (fM * em) * Ms.m;
    const float asx = (fX * exs) * Xs.sx;
    const float alx = (fX * exl) * Xs.lx;
    const float asy = (fY * eys) * Ys.sy;
    const float aly = (fY * eyl) * Ys.ly;
    Cell c;
    float b;
    b = t.mm * am;
    b = __builtin_fmaf(t.msx, asx, b);
    b = __builtin_fmaf(t.mlx, alx, b);
    b = __builtin_fmaf(t.msy, asy, b);
    b = __builtin_fmaf(t.mly, aly, b);
    c.m = b;
    b = t.sxm * am;
    b = __builtin_fmaf(t.sxsx, asx, b);
#ifndef NPR_NO_SHORT_SWITCH
    b = __builtin_fmaf(t.sxsy, asy, b);
#endif
    c.sx = b;
    b = t.sym * am;
    b = __builtin_fmaf(t.sysy, asy, b);
#ifndef NPR_NO_SHORT_SWITCH
    b = __builtin_fmaf(t.sysx, asx, b);
#endif
    c.sy = b;
    b = t.lxm * am;
    b = __builtin_fmaf(t.lxlx, alx, b);
    c.lx = b;
    b = t.lym * am;
    b = __builtin_fmaf(t.lyly, aly, b);
    c.ly = b;
    settle<NORM>(c, eref);
    return c;
}

// the rule evaluated at run time (wave-uniform `norm`): same bits as the static versions
__device__ __forceinline__ Cell fwd_cell_dyn(bool norm, const Trans &t, const Cell &L, const Cell &M, const Cell &U, float em,
                                             float exs, float exl, float eys, float eyl) {
    Cell c = fwd_cell<false>(t, L, M, U, em, exs, exl, eys, eyl);
    if (norm) normalise(c, c.e);
    return c;
}
__device__ __forceinline__ Cell bwd_cell_dyn(bool norm, const Trans &t, const Cell &Ms, const Cell &Xs, const Cell &Ys, float em,
                                             float exs, float exl, float eys, float eyl) {
    Cell c = bwd_cell<false>(t, Ms, Xs, Ys, em, exs, exl, eys, eyl);
    if (norm) normalise(c, c.e);
    return c;
}

__device__ __forceinline__ float dot5(const float *w, const Cell &c) {
    float a = w[0] * c.m;
    a = __builtin_fmaf(w[1], c.sx, a);
    a = __builtin_fmaf(w[2], c.sy, a);
    a = __builtin_fmaf(w[3], c.lx, a);
    a = __builtin_fmaf(w[4], c.ly, a);
    return a;
}

// posterior match probability of a cell: (Fv*Bv) * 2^(eF+eB-eTot) * (1/totMant)
__device__ __forceinline__ float posterior(float fv, int fe, float bv, int be, int tot_e, float inv_tot) {
    int s = fe + be - tot_e;
    s = min(max(s, -200), 200);
    return __builtin_ldexpf(fv * bv, s) * inv_tot;
}

}  // namespace npr
