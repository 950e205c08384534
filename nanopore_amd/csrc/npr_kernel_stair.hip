// npr_kernel_stair.hip -- k_dp_stair<R>: the register-resident ("systolic") DP kernel for staircase bands.
//
// Same recurrences and the same per-cell arithmetic (npr_cell.h) as k_dp_generic -- cactus_realign's banded
// five-state forward / backward / posterior pass, SURVEY.md 8a rows a5.3-a5.5, reference call sites
// nanopore/analyses/utils.py:587, alignmentUncertainty.py:41, marginAlignSnpCaller.py:136-146 -- for the
// bands every fixed-width configuration (BASELINE.json "band=100/200") and every anchor stripe produces:
// consecutive anti-diagonals whose first in-band x-y differs by exactly +-1 and that hold at most 64*R cells.
//
// Mapping (one read per 64-lane wavefront, no LDS traffic in the recurrence, no MFMA):
//   * cell j of an anti-diagonal lives in lane j / R, register j % R (blocked), so of the R neighbours a
//     step needs only ONE per state crosses a lane boundary: a single DPP wave_shl:1 / wave_shr:1 move;
//   * the two previous anti-diagonals stay in VGPRs (12*R registers each);
//   * the reference streams through the wavefront towards lower lanes on x-steps and the read towards
//     higher lanes on y-steps (one DPP move + one v_readlane injection per step), fed by 64-base blocks
//     prefetched a block ahead: no per-cell sequence loads;
//   * HMM tables in LDS (emission look-ups), transitions in SGPRs;
//   * forward match-state values stream to the wavefront's HBM scratch as R-wide vector stores and stream
//     back one anti-diagonal ahead of use in the backward sweep;
//   * band rows are read through the scalar cache (constant address space), one anti-diagonal ahead.
#include <hip/hip_runtime.h>

#include "npr_cell.h"
#include "npr_device.h"

namespace npr {

namespace {

constexpr int WAVE = 64;
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

// lane l <- lane l+1 (lane 63 keeps `edge`);  lane l <- lane l-1 (lane 0 keeps `edge`)
__device__ __forceinline__ int dpp_from_above(int v, int edge) {
    return __builtin_amdgcn_update_dpp(edge, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ int dpp_from_below(int v, int edge) {
    return __builtin_amdgcn_update_dpp(edge, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ float dppf_from_above(float v, float edge) {
    return __builtin_bit_cast(float, dpp_from_above(__builtin_bit_cast(int, v), __builtin_bit_cast(int, edge)));
}
__device__ __forceinline__ float dppf_from_below(float v, float edge) {
    return __builtin_bit_cast(float, dpp_from_below(__builtin_bit_cast(int, v), __builtin_bit_cast(int, edge)));
}

template <int R>
struct Diag {  // one anti-diagonal in registers: cell j = R*lane + r
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
    o.c[R - 1].m = dppf_from_above(in.c[0].m, 0.f);
    o.c[R - 1].sx = dppf_from_above(in.c[0].sx, 0.f);
    o.c[R - 1].sy = dppf_from_above(in.c[0].sy, 0.f);
    o.c[R - 1].lx = dppf_from_above(in.c[0].lx, 0.f);
    o.c[R - 1].ly = dppf_from_above(in.c[0].ly, 0.f);
    o.c[R - 1].e = dpp_from_above(in.c[0].e, E_DEAD);
    return o;
}
// out[j] = in[j-1]
template <int R>
__device__ __forceinline__ Diag<R> shift_down(const Diag<R> &in) {
    Diag<R> o;
#pragma unroll
    for (int r = 1; r < R; ++r) o.c[r] = in.c[r - 1];
    o.c[0].m = dppf_from_below(in.c[R - 1].m, 0.f);
    o.c[0].sx = dppf_from_below(in.c[R - 1].sx, 0.f);
    o.c[0].sy = dppf_from_below(in.c[R - 1].sy, 0.f);
    o.c[0].lx = dppf_from_below(in.c[R - 1].lx, 0.f);
    o.c[0].ly = dppf_from_below(in.c[R - 1].ly, 0.f);
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
    int off = DIR * (idx - f.base);
    if (off >= 64) {  // uniform
        f.cur = f.nxt;
        f.base += DIR * 64;
        f.nxt = base4(seq, len, f.base + DIR * (64 + lane));
        off -= 64;
    }
    return __builtin_amdgcn_readlane(f.cur, off);
}

struct BandRow {
    int lo, n;
    uint32_t co;
};

typedef const __attribute__((address_space(4))) int32_t *cptr32;

// Everything wave-uniform a step needs.
struct StepEnv {
    const DevModel *mdl;
    const char *ltab;
    Trans tr;
    const uint8_t *X, *Y;
    int lX, lY;
    int lane;
};

// A cell outside the band keeps whatever mantissas the arithmetic produced and only gets the dead exponent:
// every consumer multiplies it by scale2(E_DEAD - eref) = 0, so the mantissas never matter.
__device__ __forceinline__ void kill_outside(Cell &c, int j, int n) {
    if (j >= n) c.e = E_DEAD;
}

template <int R>
__device__ __forceinline__ void emissions(const StepEnv &E, const Bases<R> &bx, const Bases<R> &by, int r, float &em,
                                          float &exs, float &exl, float &eys, float &eyl) {
    constexpr int OFF_EM = offsetof(DevModel, em), OFF_EX = offsetof(DevModel, ex), OFF_EY = offsetof(DevModel, ey);
    em = *reinterpret_cast<const float *>(E.ltab + OFF_EM + 5 * bx.b[r] + by.b[r]);
    exs = *reinterpret_cast<const float *>(E.ltab + OFF_EX + 20 + bx.b[r]);
    exl = *reinterpret_cast<const float *>(E.ltab + OFF_EX + 60 + bx.b[r]);
    eys = *reinterpret_cast<const float *>(E.ltab + OFF_EY + 40 + by.b[r]);
    eyl = *reinterpret_cast<const float *>(E.ltab + OFF_EY + 80 + by.b[r]);
}

// One forward anti-diagonal.  `io` holds diagonal d-2 on entry and diagonal d on exit; `p1` holds d-1.  The two
// register sets swap roles every step (the caller unrolls by two), so no diagonal is ever copied.
template <int R>
__device__ __forceinline__ void fwd_step(const StepEnv &E, Diag<R> &io, const Diag<R> &p1, Bases<R> &cX, Bases<R> &cY,
                                         Feed &fx, Feed &fy, int d, int lo, int n, int lo1, int lo2, const int (&jr)[R]) {
    const int x0 = (d + lo) >> 1, y0 = (d - lo) >> 1;
    const int s = lo - lo1;          // +1: x-step, -1: y-step
    const int sm = (lo - lo2) >> 1;  // index shift into the d-2 frame: -1, 0, +1
    if (sm > 0) {
        io = shift_up<R>(io);
    } else if (sm < 0) {
        io = shift_down<R>(io);
    }
    if (s > 0) {
        bases_up<R>(cX, feed_get<+1>(fx, E.X, E.lX, x0 + 64 * R - 2, E.lane));
        const Diag<R> U = shift_up<R>(p1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float em, exs, exl, eys, eyl;
            emissions<R>(E, cX, cY, r, em, exs, exl, eys, eyl);
            Cell c = fwd_cell(E.tr, p1.c[r], io.c[r], U.c[r], em, exs, exl, eys, eyl);
            kill_outside(c, jr[r], n);
            io.c[r] = c;
        }
    } else {
        bases_down<R>(cY, feed_get<+1>(fy, E.Y, E.lY, y0 - 1, E.lane));
        const Diag<R> L = shift_down<R>(p1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float em, exs, exl, eys, eyl;
            emissions<R>(E, cX, cY, r, em, exs, exl, eys, eyl);
            Cell c = fwd_cell(E.tr, L.c[r], io.c[r], p1.c[r], em, exs, exl, eys, eyl);
            kill_outside(c, jr[r], n);
            io.c[r] = c;
        }
    }
}

template <int R>
__device__ __forceinline__ void store_row(float *Fv, int32_t *Fe, const Diag<R> &C, uint32_t co, int n, int lane) {
    if (R * lane < n) {
        if constexpr (R == 1) {
            Fv[co + lane] = C.c[0].m;
            Fe[co + lane] = C.c[0].e;
        } else if constexpr (R == 2) {
            *reinterpret_cast<float2 *>(Fv + co + 2 * lane) = make_float2(C.c[0].m, C.c[1].m);
            *reinterpret_cast<int2 *>(Fe + co + 2 * lane) = make_int2(C.c[0].e, C.c[1].e);
        } else {
            *reinterpret_cast<float4 *>(Fv + co + 4 * lane) = make_float4(C.c[0].m, C.c[1].m, C.c[2].m, C.c[3].m);
            *reinterpret_cast<int4 *>(Fe + co + 4 * lane) = make_int4(C.c[0].e, C.c[1].e, C.c[2].e, C.c[3].e);
        }
    }
}

template <int R>
struct FRow {  // forward match values of one anti-diagonal
    float v[R];
    int e[R];
};

template <int R>
__device__ __forceinline__ void load_row(const float *Fv, const int32_t *Fe, FRow<R> &f, uint32_t co, int n, int lane) {
#pragma unroll
    for (int r = 0; r < R; ++r) f.v[r] = 0.f, f.e[r] = E_DEAD;
    if (R * lane < n) {
        if constexpr (R == 1) {
            f.v[0] = Fv[co + lane];
            f.e[0] = Fe[co + lane];
        } else if constexpr (R == 2) {
            const float2 q = *reinterpret_cast<const float2 *>(Fv + co + 2 * lane);
            const int2 g = *reinterpret_cast<const int2 *>(Fe + co + 2 * lane);
            f.v[0] = q.x, f.v[1] = q.y, f.e[0] = g.x, f.e[1] = g.y;
        } else {
            const float4 q = *reinterpret_cast<const float4 *>(Fv + co + 4 * lane);
            const int4 g = *reinterpret_cast<const int4 *>(Fe + co + 4 * lane);
            f.v[0] = q.x, f.v[1] = q.y, f.v[2] = q.z, f.v[3] = q.w;
            f.e[0] = g.x, f.e[1] = g.y, f.e[2] = g.z, f.e[3] = g.w;
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
__device__ __forceinline__ void emit_pairs(const PairSink &S, const Diag<R> &B, const FRow<R> &f, int d, int lo, int n,
                                           int tot_e, float inv_tot, const int (&jr)[R], int lane, int &cnt) {
    const int x0 = (d + lo) >> 1, y0 = (d - lo) >> 1;
    float p[R];
    bool any = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        p[r] = posterior(f.v[r], f.e[r], B.c[r].m, B.c[r].e, tot_e, inv_tot);
        any |= (p[r] >= S.threshold) && (jr[r] < n);
    }
    if (d >= 2 && __ballot(any)) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool hit = (p[r] >= S.threshold) && (jr[r] < n);
            const unsigned long long mask = __ballot(hit);
            if (mask) {
                const int slot = cnt + __popcll(mask & ((1ull << lane) - 1ull));
                if (hit && slot < S.cap) {
                    S.px[S.off + slot] = x0 + jr[r] - 1 + S.xs;
                    S.py[S.off + slot] = y0 - jr[r] - 1 + S.ys;
                    S.pp[S.off + slot] = p[r];
                }
                cnt += __popcll(mask);
            }
        }
    }
}

// One backward anti-diagonal.  `io` holds diagonal d+2 on entry and d on exit; `s1` holds d+1; lo1/lo2 are the
// first x-y of d+1 and d+2.
template <int R>
__device__ __forceinline__ void bwd_step(const StepEnv &E, Diag<R> &io, const Diag<R> &s1, Bases<R> &bX, Bases<R> &bY,
                                         Feed &fx, Feed &fy, int d, int lo, int n, int lo1, int lo2, const int (&jr)[R]) {
    const int x0 = (d + lo) >> 1, y0 = (d - lo) >> 1;
    const int s = lo1 - lo;          // the forward step d -> d+1: +1 x-step, -1 y-step
    const int sm = (lo - lo2) >> 1;  // index shift into the d+2 frame
    if (sm > 0) {
        io = shift_up<R>(io);
    } else if (sm < 0) {
        io = shift_down<R>(io);
    }
    if (s > 0) {
        // x decreased by one in every slot: X[x] moves up a slot, slot 0 takes X[x0]
        bases_down<R>(bX, feed_get<-1>(fx, E.X, E.lX, x0, E.lane));
        const Diag<R> Ys = shift_down<R>(s1);  // (x, y+1) is index j-1 on d+1; (x+1, y) keeps index j
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float em, exs, exl, eys, eyl;
            emissions<R>(E, bX, bY, r, em, exs, exl, eys, eyl);
            Cell c = bwd_cell(E.tr, io.c[r], s1.c[r], Ys.c[r], em, exs, exl, eys, eyl);
            kill_outside(c, jr[r], n);
            io.c[r] = c;
        }
    } else {
        bases_up<R>(bY, feed_get<-1>(fy, E.Y, E.lY, y0 - (64 * R - 1), E.lane));
        const Diag<R> Xs = shift_up<R>(s1);    // (x+1, y) is index j+1 on d+1; (x, y+1) keeps index j
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float em, exs, exl, eys, eyl;
            emissions<R>(E, bX, bY, r, em, exs, exl, eys, eyl);
            Cell c = bwd_cell(E.tr, io.c[r], Xs.c[r], s1.c[r], em, exs, exl, eys, eyl);
            kill_outside(c, jr[r], n);
            io.c[r] = c;
        }
    }
}

template <int R>
__global__ void __launch_bounds__(WAVE) k_dp_stair(KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lmodel = reinterpret_cast<float *>(smem);
    int *lmisc = reinterpret_cast<int *>(lmodel + MODEL_FLOATS);  // 8 ints: cell hand-off

    const int lane = threadIdx.x;
    float *const Fv = a.Fv + static_cast<int64_t>(a.slot_base + blockIdx.x) * a.slot_stride;
    int32_t *const Fe = a.Fe + static_cast<int64_t>(a.slot_base + blockIdx.x) * a.slot_stride;
    int jr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) jr[r] = R * lane + r;

    int t = blockIdx.x;
    while (t < a.ntasks) {
        const Task *tp = a.tasks + t;
        const int64_t x_off = uni64(tp->x_off), y_off = uni64(tp->y_off), band_off = uni64(tp->band_off),
                      pair_off = uni64(tp->pair_off);
        const int lX = uni(tp->lX), lY = uni(tp->lY), D = uni(tp->D), pair_cap = uni(tp->pair_cap),
                  flags = uni(tp->flags), model = uni(tp->model), xs = uni(tp->xs), ys = uni(tp->ys);
        // band rows through the scalar cache: these arrays are never written by the kernel
        cptr32 blo = (cptr32)(a.lo + band_off);
        cptr32 bn = (cptr32)(a.n + band_off);
        cptr32 bco = (cptr32)(reinterpret_cast<const int32_t *>(a.coff) + band_off);
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
            if constexpr (R >= 2) {
                tr.mm = unif(tr.mm), tr.sxm = unif(tr.sxm), tr.sym = unif(tr.sym), tr.lxm = unif(tr.lxm), tr.lym = unif(tr.lym);
                tr.msx = unif(tr.msx), tr.sxsx = unif(tr.sxsx), tr.sysx = unif(tr.sysx);
                tr.msy = unif(tr.msy), tr.sysy = unif(tr.sysy), tr.sxsy = unif(tr.sxsy);
                tr.mlx = unif(tr.mlx), tr.lxlx = unif(tr.lxlx), tr.mly = unif(tr.mly), tr.lyly = unif(tr.lyly);
            }
            E.tr = tr;
        }
        const DevModel *mdl = E.mdl;

        // =============================== forward ===============================
        // A holds the even anti-diagonals, B the odd ones.
        Diag<R> A = dead_diag<R>(), B = dead_diag<R>();
        Bases<R> cX, cY;  // X[x-1]*4 and Y[y-1]*4 of every slot
#pragma unroll
        for (int r = 0; r < R; ++r) {
            cX.b[r] = base4(E.X, lX, jr[r] - 1);  // d = 0: x0 = 0, y0 = 0
            cY.b[r] = base4(E.Y, lY, -jr[r] - 1);
        }
        Feed fx, fy;
        feed_init<+1>(fx, E.X, lX, 64 * R - 1, lane);  // first x-step injects X[1 + 64R - 2]
        feed_init<+1>(fy, E.Y, lY, 0, lane);           // first y-step injects Y[0]
        BandRow r0{blo[0], bn[0], static_cast<uint32_t>(bco[0])};
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (jr[r] == 0) {
                Cell c;
                c.m = mdl->start[rs * 5 + 0], c.sx = mdl->start[rs * 5 + 1], c.sy = mdl->start[rs * 5 + 2];
                c.lx = mdl->start[rs * 5 + 3], c.ly = mdl->start[rs * 5 + 4];
                normalise(c, 0);
                A.c[r] = c;
            }
        store_row<R>(Fv, Fe, A, r0.co, r0.n, lane);
        int lo1 = r0.lo, lo2 = r0.lo;  // first x-y of d-1 and d-2
        BandRow nx{0, 0, 0};
        if (D >= 1) nx = BandRow{blo[1], bn[1], static_cast<uint32_t>(bco[1])};
        int d = 1;
        for (; d + 1 <= D; d += 2) {
            BandRow cur = nx;
            nx = BandRow{blo[d + 1], bn[d + 1], static_cast<uint32_t>(bco[d + 1])};  // one row ahead
            fwd_step<R>(E, B, A, cX, cY, fx, fy, d, cur.lo, cur.n, lo1, lo2, jr);
            store_row<R>(Fv, Fe, B, cur.co, cur.n, lane);
            lo2 = lo1, lo1 = cur.lo;
            cur = nx;
            if (d + 2 <= D) nx = BandRow{blo[d + 2], bn[d + 2], static_cast<uint32_t>(bco[d + 2])};
            fwd_step<R>(E, A, B, cX, cY, fx, fy, d + 1, cur.lo, cur.n, lo1, lo2, jr);
            store_row<R>(Fv, Fe, A, cur.co, cur.n, lane);
            lo2 = lo1, lo1 = cur.lo;
        }
        if (d <= D) {  // D odd: one more step, into B
            fwd_step<R>(E, B, A, cX, cY, fx, fy, d, nx.lo, nx.n, lo1, lo2, jr);
            store_row<R>(Fv, Fe, B, nx.co, nx.n, lane);
            lo2 = lo1, lo1 = nx.lo;
        }
        // total probability at the end corner: cell j = (lX - lY - lo_D) / 2 of the last diagonal
        {
            const int je = (lX - lY - lo1) >> 1;
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
            // P holds diagonals D, D-2, ...; Q holds D-1, D-3, ...
            Diag<R> P = dead_diag<R>(), Q = dead_diag<R>();
            Bases<R> bX, bY;  // X[x]*4 and Y[y]*4 of every slot
            BandRow cur{blo[D], bn[D], static_cast<uint32_t>(bco[D])};
            {
                const int x0 = (D + cur.lo) >> 1, y0 = (D - cur.lo) >> 1;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    bX.b[r] = base4(E.X, lX, x0 + jr[r]);
                    bY.b[r] = base4(E.Y, lY, y0 - jr[r]);
                    if (jr[r] < cur.n && x0 + jr[r] == lX && y0 - jr[r] == lY) {
                        Cell c;
                        c.m = mdl->end[re * 5 + 0], c.sx = mdl->end[re * 5 + 1], c.sy = mdl->end[re * 5 + 2];
                        c.lx = mdl->end[re * 5 + 3], c.ly = mdl->end[re * 5 + 4];
                        normalise(c, 0);
                        P.c[r] = c;
                    }
                }
                // first backward x-step injects X[x0 - 1] at slot 0; first y-step injects Y[y0 - 64R] on top
                feed_init<-1>(fx, E.X, lX, x0 - 1, lane);
                feed_init<-1>(fy, E.Y, lY, y0 - 64 * R, lane);
            }
            FRow<R> fa, fb;  // forward rows: fa pairs with P's diagonals, fb with Q's; loaded one diagonal ahead
            load_row<R>(Fv, Fe, fa, cur.co, cur.n, lane);
            BandRow nxt = cur;
            if (D >= 1) {
                nxt = BandRow{blo[D - 1], bn[D - 1], static_cast<uint32_t>(bco[D - 1])};
                load_row<R>(Fv, Fe, fb, nxt.co, nxt.n, lane);
            }
            emit_pairs<R>(sink, P, fa, D, cur.lo, cur.n, tot_e, inv_tot, jr, lane, cnt);
            int hi1 = cur.lo, hi2 = cur.lo;  // first x-y of d+1 and d+2
            int d2 = D - 1;
            for (; d2 - 1 >= 0; d2 -= 2) {
                cur = nxt;
                nxt = BandRow{blo[d2 - 1], bn[d2 - 1], static_cast<uint32_t>(bco[d2 - 1])};
                load_row<R>(Fv, Fe, fa, nxt.co, nxt.n, lane);  // for the step after this one
                bwd_step<R>(E, Q, P, bX, bY, fx, fy, d2, cur.lo, cur.n, hi1, hi2, jr);
                emit_pairs<R>(sink, Q, fb, d2, cur.lo, cur.n, tot_e, inv_tot, jr, lane, cnt);
                hi2 = hi1, hi1 = cur.lo;
                cur = nxt;
                if (d2 - 2 >= 0) {
                    nxt = BandRow{blo[d2 - 2], bn[d2 - 2], static_cast<uint32_t>(bco[d2 - 2])};
                    load_row<R>(Fv, Fe, fb, nxt.co, nxt.n, lane);
                }
                bwd_step<R>(E, P, Q, bX, bY, fx, fy, d2 - 1, cur.lo, cur.n, hi1, hi2, jr);
                emit_pairs<R>(sink, P, fa, d2 - 1, cur.lo, cur.n, tot_e, inv_tot, jr, lane, cnt);
                hi2 = hi1, hi1 = cur.lo;
            }
            if (d2 >= 0) {  // d2 == 0 left over (D odd): into Q
                bwd_step<R>(E, Q, P, bX, bY, fx, fy, d2, nxt.lo, nxt.n, hi1, hi2, jr);
                emit_pairs<R>(sink, Q, fb, d2, nxt.lo, nxt.n, tot_e, inv_tot, jr, lane, cnt);
            }
            // total from the backward side: cell (0,0) is slot 0 of diagonal 0 (in P when D is even)
            if (lane == 0) {
                const Cell c0 = (D & 1) ? Q.c[0] : P.c[0];
                const float raw = dot5(mdl->start + rs * 5, c0);
                if (raw > 0.f) {
                    int k;
                    out.btot_m = __builtin_frexpf(raw, &k);
                    out.btot_e = c0.e + k;
                }
            }
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

}  // namespace

size_t stair_lds_bytes() { return sizeof(float) * (MODEL_FLOATS + 8); }

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
