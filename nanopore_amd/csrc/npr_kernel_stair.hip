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

template <int R>
__global__ void __launch_bounds__(WAVE) k_dp_stair(KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lmodel = reinterpret_cast<float *>(smem);
    int *lmisc = reinterpret_cast<int *>(lmodel + MODEL_FLOATS);  // 8 ints: cell hand-off
    const char *ltab = reinterpret_cast<const char *>(lmodel);    // byte-addressed table look-ups

    const int lane = threadIdx.x;
    float *const Fv = a.Fv + static_cast<int64_t>(blockIdx.x) * a.slot_stride;
    int32_t *const Fe = a.Fe + static_cast<int64_t>(blockIdx.x) * a.slot_stride;
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
        const uint8_t *X = a.seq + x_off;
        const uint8_t *Y = a.seq + y_off;
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
        const DevModel *mdl = reinterpret_cast<const DevModel *>(lmodel);
        Trans tr = load_trans(mdl->T);
        tr.mm = unif(tr.mm), tr.sxm = unif(tr.sxm), tr.sym = unif(tr.sym), tr.lxm = unif(tr.lxm), tr.lym = unif(tr.lym);
        tr.msx = unif(tr.msx), tr.sxsx = unif(tr.sxsx), tr.sysx = unif(tr.sysx);
        tr.msy = unif(tr.msy), tr.sysy = unif(tr.sysy), tr.sxsy = unif(tr.sxsy);
        tr.mlx = unif(tr.mlx), tr.lxlx = unif(tr.lxlx), tr.mly = unif(tr.mly), tr.lyly = unif(tr.lyly);
        // byte offsets of the tables inside the staged model
        constexpr int OFF_EM = offsetof(DevModel, em), OFF_EX = offsetof(DevModel, ex), OFF_EY = offsetof(DevModel, ey);

        // =============================== forward ===============================
        Diag<R> P1 = dead_diag<R>(), P2 = dead_diag<R>();
        Bases<R> cX, cY;  // X[x-1]*4 and Y[y-1]*4 of every slot
#pragma unroll
        for (int r = 0; r < R; ++r) {
            cX.b[r] = base4(X, lX, jr[r] - 1);  // d = 0: x0 = 0, y0 = 0
            cY.b[r] = base4(Y, lY, -jr[r] - 1);
        }
        Feed fx, fy;
        feed_init<+1>(fx, X, lX, 64 * R - 1, lane);  // first x-step injects X[1 + 64R - 2]
        feed_init<+1>(fy, Y, lY, 0, lane);           // first y-step injects Y[0]
        int lo1 = 0, lo2 = 0;
        BandRow row{blo[0], bn[0], static_cast<uint32_t>(bco[0])};
        for (int d = 0; d <= D; ++d) {
            const BandRow cur = row;
            if (d < D) row = BandRow{blo[d + 1], bn[d + 1], static_cast<uint32_t>(bco[d + 1])};  // one row ahead
            const int lo = cur.lo, n = cur.n;
            const int x0 = (d + lo) >> 1, y0 = (d - lo) >> 1;
            Diag<R> C;
            if (d == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    Cell c = dead_cell();
                    if (jr[r] == 0) {
                        c.m = mdl->start[rs * 5 + 0], c.sx = mdl->start[rs * 5 + 1], c.sy = mdl->start[rs * 5 + 2];
                        c.lx = mdl->start[rs * 5 + 3], c.ly = mdl->start[rs * 5 + 4];
                        normalise(c, 0);
                    }
                    C.c[r] = c;
                }
            } else {
                const int s = lo - lo1;               // +1: x-step, -1: y-step
                const int sm = (lo - lo2) >> 1;       // shift of the d-2 frame: -1, 0, +1 (d == 1: unused, P2 dead)
                Diag<R> L, U, M;
                if (s > 0) {
                    bases_up<R>(cX, feed_get<+1>(fx, X, lX, x0 + 64 * R - 2, lane));
                    L = P1;
                    U = shift_up<R>(P1);
                } else {
                    bases_down<R>(cY, feed_get<+1>(fy, Y, lY, y0 - 1, lane));
                    U = P1;
                    L = shift_down<R>(P1);
                }
                if (sm == 0) {
                    M = P2;
                } else if (sm > 0) {
                    M = shift_up<R>(P2);
                } else {
                    M = shift_down<R>(P2);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float em = *reinterpret_cast<const float *>(ltab + OFF_EM + 5 * cX.b[r] + cY.b[r]);
                    const float exs = *reinterpret_cast<const float *>(ltab + OFF_EX + 20 + cX.b[r]);
                    const float exl = *reinterpret_cast<const float *>(ltab + OFF_EX + 60 + cX.b[r]);
                    const float eys = *reinterpret_cast<const float *>(ltab + OFF_EY + 40 + cY.b[r]);
                    const float eyl = *reinterpret_cast<const float *>(ltab + OFF_EY + 80 + cY.b[r]);
                    Cell c = fwd_cell(tr, L.c[r], M.c[r], U.c[r], em, exs, exl, eys, eyl);
                    if (jr[r] >= n) c = dead_cell();
                    C.c[r] = c;
                }
            }
            // stream the match state to HBM: R consecutive cells per lane
            if (R * lane < n) {
                if constexpr (R == 1) {
                    Fv[cur.co + lane] = C.c[0].m;
                    Fe[cur.co + lane] = C.c[0].e;
                } else if constexpr (R == 2) {
                    *reinterpret_cast<float2 *>(Fv + cur.co + 2 * lane) = make_float2(C.c[0].m, C.c[1].m);
                    *reinterpret_cast<int2 *>(Fe + cur.co + 2 * lane) = make_int2(C.c[0].e, C.c[1].e);
                } else {
                    *reinterpret_cast<float4 *>(Fv + cur.co + 4 * lane) = make_float4(C.c[0].m, C.c[1].m, C.c[2].m, C.c[3].m);
                    *reinterpret_cast<int4 *>(Fe + cur.co + 4 * lane) = make_int4(C.c[0].e, C.c[1].e, C.c[2].e, C.c[3].e);
                }
            }
            P2 = P1;
            P1 = C;
            lo2 = lo1, lo1 = lo;
        }
        // total probability at the end corner: cell j = (lX - lY - lo_D) / 2 of the last diagonal
        {
            const int je = (lX - lY - lo1) >> 1;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (jr[r] == je) {
                    const Cell &c = P1.c[r];
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
            Diag<R> S1 = dead_diag<R>(), S2 = dead_diag<R>();  // diagonals d+1 and d+2
            Bases<R> bX, bY;  // X[x]*4 and Y[y]*4 of every slot
            {
                const int x0 = (D + lo1) >> 1, y0 = (D - lo1) >> 1;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    bX.b[r] = base4(X, lX, x0 + jr[r]);
                    bY.b[r] = base4(Y, lY, y0 - jr[r]);
                }
                // first backward x-step injects X[x0 - 1] at slot 0; first y-step injects Y[y0 - 1 - (64R-1)] on top
                feed_init<-1>(fx, X, lX, x0 - 1, lane);
                feed_init<-1>(fy, Y, lY, y0 - 64 * R, lane);
            }
            int hi1 = 0, hi2 = 0;  // lo of d+1, d+2
            BandRow brow{blo[D], bn[D], static_cast<uint32_t>(bco[D])};
            // forward values of the current diagonal, loaded one diagonal ahead
            float fv[R];
            int fe[R];
            auto loadF = [&](const BandRow &w, float (&v)[R], int (&e)[R]) {
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = 0.f, e[r] = E_DEAD;
                if (R * lane < w.n) {
                    if constexpr (R == 1) {
                        v[0] = Fv[w.co + lane];
                        e[0] = Fe[w.co + lane];
                    } else if constexpr (R == 2) {
                        const float2 q = *reinterpret_cast<const float2 *>(Fv + w.co + 2 * lane);
                        const int2 g = *reinterpret_cast<const int2 *>(Fe + w.co + 2 * lane);
                        v[0] = q.x, v[1] = q.y, e[0] = g.x, e[1] = g.y;
                    } else {
                        const float4 q = *reinterpret_cast<const float4 *>(Fv + w.co + 4 * lane);
                        const int4 g = *reinterpret_cast<const int4 *>(Fe + w.co + 4 * lane);
                        v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w, e[0] = g.x, e[1] = g.y, e[2] = g.z, e[3] = g.w;
                    }
                }
            };
            loadF(brow, fv, fe);
            for (int d = D; d >= 0; --d) {
                const BandRow cur = brow;
                float fvn[R];
                int fen[R];
                if (d > 0) {
                    brow = BandRow{blo[d - 1], bn[d - 1], static_cast<uint32_t>(bco[d - 1])};
                    loadF(brow, fvn, fen);  // issued a whole diagonal ahead of its use
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) fvn[r] = 0.f, fen[r] = E_DEAD;
                }
                const int lo = cur.lo, n = cur.n;
                const int x0 = (d + lo) >> 1, y0 = (d - lo) >> 1;
                Diag<R> C;
                if (d == D) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        Cell c = dead_cell();
                        if (jr[r] < n && x0 + jr[r] == lX && y0 - jr[r] == lY) {
                            c.m = mdl->end[re * 5 + 0], c.sx = mdl->end[re * 5 + 1], c.sy = mdl->end[re * 5 + 2];
                            c.lx = mdl->end[re * 5 + 3], c.ly = mdl->end[re * 5 + 4];
                            normalise(c, 0);
                        }
                        C.c[r] = c;
                    }
                } else {
                    const int s = hi1 - lo;            // the forward step d -> d+1: +1 x-step, -1 y-step
                    const int sm = (lo - hi2) >> 1;    // index shift into the d+2 frame (d+2 > D: S2 dead)
                    Diag<R> Xs, Ys, Ms;
                    if (s > 0) {
                        // x decreased by one for every slot: X[x] moves up a slot, slot 0 takes X[x0]
                        bases_down<R>(bX, feed_get<-1>(fx, X, lX, x0, lane));
                        Xs = S1;               // (x+1, y) has the same index on d+1
                        Ys = shift_down<R>(S1);  // (x, y+1) is index j-1 on d+1
                    } else {
                        bases_up<R>(bY, feed_get<-1>(fy, Y, lY, y0 - (64 * R - 1), lane));
                        Xs = shift_up<R>(S1);
                        Ys = S1;
                    }
                    if (d + 2 > D || sm == 0) {
                        Ms = S2;
                    } else if (sm > 0) {
                        Ms = shift_up<R>(S2);
                    } else {
                        Ms = shift_down<R>(S2);
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float em = *reinterpret_cast<const float *>(ltab + OFF_EM + 5 * bX.b[r] + bY.b[r]);
                        const float exs = *reinterpret_cast<const float *>(ltab + OFF_EX + 20 + bX.b[r]);
                        const float exl = *reinterpret_cast<const float *>(ltab + OFF_EX + 60 + bX.b[r]);
                        const float eys = *reinterpret_cast<const float *>(ltab + OFF_EY + 40 + bY.b[r]);
                        const float eyl = *reinterpret_cast<const float *>(ltab + OFF_EY + 80 + bY.b[r]);
                        Cell c = bwd_cell(tr, Ms.c[r], Xs.c[r], Ys.c[r], em, exs, exl, eys, eyl);
                        if (jr[r] >= n) c = dead_cell();
                        C.c[r] = c;
                    }
                }
                // posteriors of this diagonal
                float p[R];
                bool anyhit = false;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    p[r] = posterior(fv[r], fe[r], C.c[r].m, C.c[r].e, tot_e, inv_tot);
                    // pairs need x >= 1 and y >= 1: from d = 2 on, the forward match value is zero elsewhere
                    // (d = 0 is the start cell, whose match state holds the start probability)
                    anyhit |= (p[r] >= a.threshold) && (jr[r] < n) && (d >= 2);
                }
                if (__ballot(anyhit)) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const bool hit = (p[r] >= a.threshold) && (jr[r] < n);
                        const unsigned long long mask = __ballot(hit);
                        if (mask) {
                            const int slot = cnt + __popcll(mask & ((1ull << lane) - 1ull));
                            if (hit && slot < pair_cap) {
                                a.px[pair_off + slot] = x0 + jr[r] - 1 + xs;
                                a.py[pair_off + slot] = y0 - jr[r] - 1 + ys;
                                a.pp[pair_off + slot] = p[r];
                            }
                            cnt += __popcll(mask);
                        }
                    }
                }
                S2 = S1;
                S1 = C;
                hi2 = hi1, hi1 = lo;
#pragma unroll
                for (int r = 0; r < R; ++r) fv[r] = fvn[r], fe[r] = fen[r];
            }
            // total from the backward side: cell (0,0) is slot 0 of diagonal 0
            if (lane == 0) {
                const float raw = dot5(mdl->start + rs * 5, S1.c[0]);
                if (raw > 0.f) {
                    int k;
                    out.btot_m = __builtin_frexpf(raw, &k);
                    out.btot_e = S1.c[0].e + k;
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
