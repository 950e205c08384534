// npr_kernels.hip -- HIP kernels of libnprealign for gfx950 (MI355X, CDNA4).
//
// k_dp_generic: the banded five-state pair-HMM forward / backward / posterior pass of cactus_realign
// (reference call sites nanopore/analyses/utils.py:587,
// alignmentUncertainty.py:41, marginAlignSnpCaller.py:136-146) for ARBITRARY bands.
//   * one DP problem (task) per workgroup of 1, 4 or 8 wavefronts (the wider the band, the more wavefronts, so
//     that big tasks -- whose forward scratch limits how many can be resident -- still fill the chip); threads
//     stride over the cells of an anti-diagonal;
//   * the two previous anti-diagonals live in an LDS ring (SoA, conflict-free), HMM tables in LDS;
//   * persistent wavefronts pull tasks from an atomic queue (tasks are sorted longest-first);
//   * forward and backward of a task are fused in one launch: the forward match-state values are
//     streamed to a per-wavefront HBM scratch region (coalesced) and streamed back by the backward sweep,
//     which emits the sparse posterior list by ballot/popcount compaction.
// No MFMA: this is a recurrence, not a contraction.
#include <hip/hip_runtime.h>

#include "npr_cell.h"
#include "npr_device.h"

namespace npr {

namespace {

constexpr int WAVE = 64;
constexpr int MODEL_FLOATS = sizeof(DevModel) / sizeof(float);

// Wave-uniform values are forced into SGPRs explicitly.  Left to itself hipcc keeps values loaded through
// the vector path (the task descriptor, band rows) in VGPRs and wraps their scalar uses in waterfall loops
// -- around the whole task body here, which both serialises and (ROCm 7.2) hangs the kernel.
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uniformf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int64_t uniform64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32));
    return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}

// the task descriptor with every field in SGPRs
struct UTask {
    int64_t x_off, y_off, band_off, pair_off;
    int lX, lY, D, pair_cap, flags, model, xs, ys;
};
__device__ __forceinline__ UTask load_task(const Task *p) {
    UTask u;
    u.x_off = uniform64(p->x_off), u.y_off = uniform64(p->y_off);
    u.band_off = uniform64(p->band_off), u.pair_off = uniform64(p->pair_off);
    u.lX = uniform(p->lX), u.lY = uniform(p->lY), u.D = uniform(p->D), u.pair_cap = uniform(p->pair_cap);
    u.flags = uniform(p->flags), u.model = uniform(p->model), u.xs = uniform(p->xs), u.ys = uniform(p->ys);
    return u;
}

struct Ring {
    float *base;
    int wcap;
    __device__ __forceinline__ float *comp(int slot, int c) const { return base + (slot * 6 + c) * wcap; }
    __device__ __forceinline__ Cell get(int slot, int j) const {
        Cell c;
        c.m = comp(slot, 0)[j];
        c.sx = comp(slot, 1)[j];
        c.sy = comp(slot, 2)[j];
        c.lx = comp(slot, 3)[j];
        c.ly = comp(slot, 4)[j];
        c.e = reinterpret_cast<const int *>(comp(slot, 5))[j];
        return c;
    }
    __device__ __forceinline__ void put(int slot, int j, const Cell &c) const {
        comp(slot, 0)[j] = c.m;
        comp(slot, 1)[j] = c.sx;
        comp(slot, 2)[j] = c.sy;
        comp(slot, 3)[j] = c.lx;
        comp(slot, 4)[j] = c.ly;
        reinterpret_cast<int *>(comp(slot, 5))[j] = c.e;
    }
};

// index of the cell with in-diagonal coordinate xmy on a diagonal whose band is (lo, n); -1 if outside
__device__ __forceinline__ int band_index(int xmy, int lo, int n) {
    const int t = xmy - lo;
    const int j = t >> 1;
    return (t >= 0 && j < n) ? j : -1;
}

// GLOBAL_RING: the three-diagonal ring lives in a per-wavefront HBM/L2 region instead of LDS, for the rare
// anti-diagonals wider than the 160 KiB of LDS can hold (unanchored rectangles up to
// splitMatrixBiggerThanThis = 3000 cells across).  One wavefront owns the region, and __syncthreads() between
// anti-diagonals carries the workgroup-scope release/acquire that orders its own stores and loads.
// EM: Baum-Welch E-step (cactus_realign --outputExpectations, summed by
// cactus_expectationMaximisation at nanopore/analyses/utils.py:509-528).  The forward sweep keeps all five
// states per cell in HBM; the backward sweep, after finishing a cell, adds the posterior probability of every
// transition INTO that cell to 15 per-lane accumulators and the emitted symbols' posterior to per-lane LDS bins;
// the wavefront reduces them at the end of the task and adds them to the model's global counts (fp64 atomics).
template <bool DENSE, bool GLOBAL_RING, bool EM>
__global__ void __launch_bounds__(512) k_dp_generic(KernelArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ring ring{GLOBAL_RING ? a.ring + static_cast<int64_t>(blockIdx.x) * 18 * a.wcap : reinterpret_cast<float *>(smem), a.wcap};
    float *lmodel = reinterpret_cast<float *>(smem) + (GLOBAL_RING ? 0 : 18 * a.wcap);
    int *lmisc = reinterpret_cast<int *>(lmodel + MODEL_FLOATS);  // [0..1] totals hand-off
    float *lbins = reinterpret_cast<float *>(lmisc + 4);          // EM only: (EM_BINS + 15) rows of 64 lanes
    float *const Fx = EM ? a.Fx + static_cast<int64_t>(blockIdx.x) * 4 * a.slot_stride : nullptr;

    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int nthreads = blockDim.x;  // 64, 256 or 512 (the E-step variant always runs 64)
    float *const Fv = reinterpret_cast<float *>(a.F + static_cast<int64_t>(a.slot_base + blockIdx.x) * a.slot_stride * 8);
    int32_t *const Fe = reinterpret_cast<int32_t *>(Fv + a.slot_stride);

    // The first task of every wavefront is static (t = blockIdx.x); later ones come from the atomic queue.
    // The loop is kept free of break/continue and every loop-carried value is an SGPR: with a divergent
    // fetch at the loop head hipcc (ROCm 7.2) structurised the task loop as a divergent loop and hung.
    int t = blockIdx.x;
    while (t < a.ntasks) {
        const UTask tk = load_task(a.tasks + t);
        const int lX = tk.lX, lY = tk.lY, D = tk.D;
        const uint8_t *X = a.seq + tk.x_off;
        const uint8_t *Y = a.seq + tk.y_off;
        const int32_t *blo = a.lo + tk.band_off;
        const int32_t *bn = a.n + tk.band_off;
        const uint32_t *bco = a.coff + tk.band_off;
        const int rs = tk.flags & 1, re = (tk.flags >> 1) & 1;

        __syncthreads();  // previous task's LDS reads are done
        {
            const float *gm = reinterpret_cast<const float *>(a.models + tk.model);
            for (int i = tid; i < MODEL_FLOATS; i += nthreads) lmodel[i] = gm[i];
            if (tid == 0) lmisc[2] = 0;  // posterior pair counter of the workgroup
        }
        __syncthreads();
        const DevModel *mdl = reinterpret_cast<const DevModel *>(lmodel);
        const Trans tr = load_trans(mdl->T);

        // ------------------------------- forward -------------------------------
        int lo1 = 0, n1 = 0, lo2 = 0, n2 = 0;  // bands of d-1 and d-2
        for (int d = 0; d <= D; ++d) {
            const int lo = uniform(blo[d]), n = uniform(bn[d]);
            const uint32_t co = static_cast<uint32_t>(uniform(static_cast<int>(bco[d])));
            const int cur = d % 3, s1 = (d + 2) % 3, s2 = (d + 1) % 3;
            for (int j = tid; j < n; j += nthreads) {
                const int xmy = lo + 2 * j;
                const int x = (d + xmy) >> 1, y = (d - xmy) >> 1;
                Cell c = dead_cell();
                if (x >= 0 && y >= 0 && x <= lX && y <= lY) {
                    if (d == 0) {
                        c.m = mdl->start[rs * 5 + 0];
                        c.sx = mdl->start[rs * 5 + 1];
                        c.sy = mdl->start[rs * 5 + 2];
                        c.lx = mdl->start[rs * 5 + 3];
                        c.ly = mdl->start[rs * 5 + 4];
                        normalise(c, 0);
                    } else {
                        const int jL = (x > 0 && d >= 1) ? band_index(xmy - 1, lo1, n1) : -1;
                        const int jU = (y > 0 && d >= 1) ? band_index(xmy + 1, lo1, n1) : -1;
                        const int jM = (x > 0 && y > 0 && d >= 2) ? band_index(xmy, lo2, n2) : -1;
                        const Cell L = jL >= 0 ? ring.get(s1, jL) : dead_cell();
                        const Cell U = jU >= 0 ? ring.get(s1, jU) : dead_cell();
                        const Cell M = jM >= 0 ? ring.get(s2, jM) : dead_cell();
                        const int cx = x > 0 ? X[x - 1] : 4, cy = y > 0 ? Y[y - 1] : 4;
                        c = fwd_cell_dyn(norm_diag(d), tr, L, M, U, mdl->em[cx * 5 + cy], mdl->ex[5 + cx], mdl->ex[15 + cx],
                                     mdl->ey[10 + cy], mdl->ey[20 + cy]);
                    }
                }
                ring.put(cur, j, c);
                Fv[co + j] = c.m;
                Fe[co + j] = c.e;
                if (EM) {
                    Fx[co + j] = c.sx;
                    Fx[a.slot_stride + co + j] = c.sy;
                    Fx[2 * a.slot_stride + co + j] = c.lx;
                    Fx[3 * a.slot_stride + co + j] = c.ly;
                }
            }
            __syncthreads();
            lo2 = lo1, n2 = n1, lo1 = lo, n1 = n;
        }
        // total probability at the end corner (lX, lY) of anti-diagonal D
        if (tid == 0) {
            float tm = 0.f;
            int te = E_DEAD;
            const int je = band_index(lX - lY, lo1, n1);
            if (je >= 0) {
                const Cell c = ring.get(D % 3, je);
                const float raw = dot5(mdl->end + re * 5, c);
                if (raw > 0.f) {
                    int k;
                    tm = __builtin_frexpf(raw, &k);
                    te = c.e + k;
                }
            }
            reinterpret_cast<float *>(lmisc)[0] = tm;
            lmisc[1] = te;
        }
        __syncthreads();
        const float tot_m = uniformf(reinterpret_cast<float *>(lmisc)[0]);
        const int tot_e = uniform(lmisc[1]);
        __syncthreads();

        TaskOut out;
        out.tot_m = tot_m, out.tot_e = tot_e, out.btot_m = 0.f, out.btot_e = E_DEAD, out.npairs = 0;
        out.status = NPR_OK;
        const bool alive = tot_m > 0.f;  // wave-uniform (SGPR compare)
        if (!alive) out.status = NPR_ERR_ZERO_PROB;

        // ------------------------------- backward + posteriors -------------------------------
        const float inv_tot = 1.0f / tot_m;
        lo1 = n1 = lo2 = n2 = 0;  // bands of d+1 and d+2
        float acc[15];            // EM: expected transition counts of this lane
        if (EM) {
#pragma unroll
            for (int i = 0; i < 15; ++i) acc[i] = 0.f;
            for (int i = 0; i < EM_BINS; ++i) lbins[i * WAVE + lane] = 0.f;  // (the E-step runs one wavefront per task)
        }
        for (int d = alive ? D : -1; d >= 0; --d) {
            const int lo = uniform(blo[d]), n = uniform(bn[d]);
            const uint32_t co = static_cast<uint32_t>(uniform(static_cast<int>(bco[d])));
            const int cur = d % 3, s1 = (d + 1) % 3, s2 = (d + 2) % 3;
            // EM: band rows of the predecessors' diagonals
            int lom1 = 0, nm1 = 0, lom2 = 0, nm2 = 0;
            uint32_t com1 = 0, com2 = 0;
            if (EM && d >= 1) {
                lom1 = uniform(blo[d - 1]), nm1 = uniform(bn[d - 1]);
                com1 = static_cast<uint32_t>(uniform(static_cast<int>(bco[d - 1])));
            }
            if (EM && d >= 2) {
                lom2 = uniform(blo[d - 2]), nm2 = uniform(bn[d - 2]);
                com2 = static_cast<uint32_t>(uniform(static_cast<int>(bco[d - 2])));
            }
            for (int j0 = 0; j0 < n; j0 += nthreads) {
                const int j = j0 + tid;
                bool hit = false;
                float p = 0.f;
                int x = 0, y = 0;
                if (j < n) {
                    const int xmy = lo + 2 * j;
                    x = (d + xmy) >> 1, y = (d - xmy) >> 1;
                    Cell c = dead_cell();
                    if (x >= 0 && y >= 0 && x <= lX && y <= lY) {
                        if (d == D) {
                            c.m = mdl->end[re * 5 + 0];
                            c.sx = mdl->end[re * 5 + 1];
                            c.sy = mdl->end[re * 5 + 2];
                            c.lx = mdl->end[re * 5 + 3];
                            c.ly = mdl->end[re * 5 + 4];
                            normalise(c, 0);
                        } else {
                            const int jX = (x < lX) ? band_index(xmy + 1, lo1, n1) : -1;
                            const int jY = (y < lY) ? band_index(xmy - 1, lo1, n1) : -1;
                            const int jM = (x < lX && y < lY && d + 2 <= D) ? band_index(xmy, lo2, n2) : -1;
                            const Cell Xs = jX >= 0 ? ring.get(s1, jX) : dead_cell();
                            const Cell Ys = jY >= 0 ? ring.get(s1, jY) : dead_cell();
                            const Cell Ms = jM >= 0 ? ring.get(s2, jM) : dead_cell();
                            const int cx = x < lX ? X[x] : 4, cy = y < lY ? Y[y] : 4;
                            c = bwd_cell_dyn(norm_diag(d), tr, Ms, Xs, Ys, mdl->em[cx * 5 + cy], mdl->ex[5 + cx], mdl->ex[15 + cx],
                                         mdl->ey[10 + cy], mdl->ey[20 + cy]);
                        }
                        if (x >= 1 && y >= 1 && !EM) {
                            p = posterior(Fv[co + j], Fe[co + j], c.m, c.e, tot_e, inv_tot);
                            hit = p >= a.threshold;
                        }
                        if (EM && d >= 1 && c.e != E_DEAD) {
                            const int ss = static_cast<int>(a.slot_stride);
                            const int ex = x > 0 ? X[x - 1] : 4, ey = y > 0 ? Y[y - 1] : 4;  // bases consumed INTO this cell
                            const int jM = (x > 0 && y > 0 && d >= 2) ? band_index(xmy, lom2, nm2) : -1;
                            const int jL = (x > 0) ? band_index(xmy - 1, lom1, nm1) : -1;
                            const int jU = (y > 0) ? band_index(xmy + 1, lom1, nm1) : -1;
                            if (jM >= 0) {
                                const uint32_t q = com2 + jM;
                                const int s = min(max(Fe[q] + c.e - tot_e, -200), 200);
                                const float w = __builtin_ldexpf(mdl->em[ex * 5 + ey] * c.m * inv_tot, s);
                                const float t0 = Fv[q] * tr.mm * w, t1 = Fx[q] * tr.sxm * w, t2 = Fx[ss + q] * tr.sym * w,
                                            t3 = Fx[2 * ss + q] * tr.lxm * w, t4 = Fx[3 * ss + q] * tr.lym * w;
                                acc[0] += t0, acc[1] += t1, acc[2] += t2, acc[3] += t3, acc[4] += t4;
                                if (ex < 4 && ey < 4) lbins[(ex * 4 + ey) * WAVE + lane] += (t0 + t1) + (t2 + t3) + t4;
                            }
                            if (jL >= 0) {
                                const uint32_t q = com1 + jL;
                                const int s = min(max(Fe[q] + c.e - tot_e, -200), 200);
                                const float g = __builtin_ldexpf(inv_tot, s);
                                const float ws = mdl->ex[5 + ex] * c.sx * g, wl = mdl->ex[15 + ex] * c.lx * g;
                                const float fm = Fv[q];
                                const float t0 = fm * tr.msx * ws, t1 = Fx[q] * tr.sxsx * ws, t2 = Fx[ss + q] * tr.sysx * ws;
                                const float u0 = fm * tr.mlx * wl, u1 = Fx[2 * ss + q] * tr.lxlx * wl;
                                acc[5] += t0, acc[6] += t1, acc[7] += t2, acc[8] += u0, acc[9] += u1;
                                if (ex < 4) {
                                    lbins[(16 + ex) * WAVE + lane] += (t0 + t1) + t2;
                                    lbins[(20 + ex) * WAVE + lane] += u0 + u1;
                                }
                            }
                            if (jU >= 0) {
                                const uint32_t q = com1 + jU;
                                const int s = min(max(Fe[q] + c.e - tot_e, -200), 200);
                                const float g = __builtin_ldexpf(inv_tot, s);
                                const float ws = mdl->ey[10 + ey] * c.sy * g, wl = mdl->ey[20 + ey] * c.ly * g;
                                const float fm = Fv[q];
                                const float t0 = fm * tr.msy * ws, t1 = Fx[ss + q] * tr.sysy * ws, t2 = Fx[q] * tr.sxsy * ws;
                                const float u0 = fm * tr.mly * wl, u1 = Fx[3 * ss + q] * tr.lyly * wl;
                                acc[10] += t0, acc[11] += t1, acc[12] += t2, acc[13] += u0, acc[14] += u1;
                                if (ey < 4) {
                                    lbins[(24 + ey) * WAVE + lane] += (t0 + t1) + t2;
                                    lbins[(28 + ey) * WAVE + lane] += u0 + u1;
                                }
                            }
                        }
                    }
                    ring.put(cur, j, c);
                    if (DENSE) {
                        a.Bv[co + j] = c.m;
                        a.Be[co + j] = c.e;
                    }
                }
                const unsigned long long mask = __ballot(hit);
                if (mask) {  // wave-uniform: one LDS atomic per wavefront reserves the slots of its hits
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&lmisc[2], __popcll(mask));
                    base = uniform(base);
                    const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
                    if (hit && slot < tk.pair_cap) {
                        a.px[tk.pair_off + slot] = x - 1 + tk.xs;
                        a.py[tk.pair_off + slot] = y - 1 + tk.ys;
                        a.pp[tk.pair_off + slot] = p;
                    }
                }
            }
            __syncthreads();
            lo2 = lo1, n2 = n1, lo1 = lo, n1 = n;
        }
        if (EM && alive) {
            // transition accumulators -> LDS rows EM_BINS .. EM_BINS+14, then one lane per row sums 64 values
#pragma unroll
            for (int i = 0; i < 15; ++i) lbins[(EM_BINS + i) * WAVE + lane] = acc[i];
            __syncthreads();
            if (lane < EM_BINS + 15) {
                double sum = 0.0;
                for (int q = 0; q < WAVE; ++q) sum += static_cast<double>(lbins[lane * WAVE + q]);
                if (lane < EM_BINS) {
                    atomicAdd(a.em_E + tk.model * EM_BINS + lane, sum);
                } else {
                    // accumulator order -> T[from*5+to]
                    const int map[15] = {0, 5, 10, 15, 20, 1, 6, 11, 3, 18, 2, 12, 7, 4, 24};
                    atomicAdd(a.em_T + tk.model * 25 + map[lane - EM_BINS], sum);
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            const int cnt = lmisc[2];
            const int j0 = alive ? band_index(0, lo1, n1) : -1;
            if (j0 >= 0) {
                const Cell c = ring.get(0, j0);
                const float raw = dot5(mdl->start + rs * 5, c);
                if (raw > 0.f) {
                    int k;
                    out.btot_m = __builtin_frexpf(raw, &k);
                    out.btot_e = c.e + k;
                }
            }
            out.npairs = cnt;
            if (cnt > tk.pair_cap) out.status = NPR_ERR_CAPACITY;
            a.outs[t] = out;
        }
        // next task: one fetch per workgroup, handed to every wavefront through LDS
        if (tid == 0) lmisc[3] = atomicAdd(a.queue, 1) + static_cast<int>(gridDim.x);
        __syncthreads();
        t = uniform(lmisc[3]);
    }
}

// gathers each task's posterior pairs into one dense buffer for a single D2H copy
__global__ void __launch_bounds__(256) k_compact(CompactArgs a) {
    for (int t = blockIdx.x; t < a.ntasks; t += gridDim.x) {
        const UTask tk = load_task(a.tasks + t);
        const int n = min(a.outs[t].npairs, tk.pair_cap);
        const int64_t dst = a.dst_off[t];
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            a.cx[dst + i] = a.px[tk.pair_off + i];
            a.cy[dst + i] = a.py[tk.pair_off + i];
            a.cp[dst + i] = a.pp[tk.pair_off + i];
        }
    }
}

}  // namespace

// wcap <= 0: global-ring variant (LDS holds only the model tables)
size_t generic_lds_bytes(int wcap) { return sizeof(float) * (18 * static_cast<size_t>(wcap > 0 ? wcap : 0) + MODEL_FLOATS + 4); }

int generic_max_wcap() {
    // 160 KiB of LDS per workgroup on gfx950
    return static_cast<int>((160 * 1024 / sizeof(float) - MODEL_FLOATS - 4) / 18) & ~3;
}

template <bool DENSE, bool GLOBAL_RING, bool EM>
static int launch_generic_t(const KernelArgs &a, int grid, int threads, size_t lds_bytes, hipStream_t s) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dp_generic<DENSE, GLOBAL_RING, EM>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
    if (e != hipSuccess) return static_cast<int>(e);
    hipLaunchKernelGGL((k_dp_generic<DENSE, GLOBAL_RING, EM>), dim3(grid), dim3(threads), lds_bytes, s, a);
    return static_cast<int>(hipGetLastError());
}

int launch_generic(const KernelArgs &a, int grid, int threads, size_t lds_bytes, bool dense, bool global_ring, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dense) return global_ring ? launch_generic_t<true, true, false>(a, grid, threads, lds_bytes, s) : launch_generic_t<true, false, false>(a, grid, threads, lds_bytes, s);
    return global_ring ? launch_generic_t<false, true, false>(a, grid, threads, lds_bytes, s) : launch_generic_t<false, false, false>(a, grid, threads, lds_bytes, s);
}

// lds_bytes must include em_extra_lds_bytes()
int launch_em(const KernelArgs &a, int grid, size_t lds_bytes, bool global_ring, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    return global_ring ? launch_generic_t<false, true, true>(a, grid, WAVE, lds_bytes, s) : launch_generic_t<false, false, true>(a, grid, WAVE, lds_bytes, s);
}

size_t em_extra_lds_bytes() { return sizeof(float) * (EM_BINS + 15) * WAVE; }

int launch_compact(const CompactArgs &a, void *stream) {
    const int grid = a.ntasks < 4096 ? (a.ntasks > 0 ? a.ntasks : 1) : 4096;
    hipLaunchKernelGGL(k_compact, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}

}  // namespace npr
