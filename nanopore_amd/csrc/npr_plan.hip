// npr_plan.hip -- the device half of batch staging.
//
// cactus_realign builds a read's band per anti-diagonal on the CPU, and so did
// npr_batch_create: 2.5e8 anti-diagonals for the default bench batch, 20 bytes each, expanded by 16 host threads and
// pushed over PCIe -- a second of staging for 0.15 s of DP.  The host now does only the O(#cigar ops) part (anchors,
// trimming, matrix splits: npr_host.cpp plan_points) and uploads each segment's chain of plan points (8 bytes per cigar
// operation); the kernels below expand them where the DP kernels read them:
//   k_plan_bands    band rows lo / n of every anti-diagonal (npr_band.h: the same function the host planner uses) and per
//                   segment the cells, the widest anti-diagonal and the generic kernel's scratch need;
//   k_plan_sched    the frame schedule of the register kernels (npr_sched.h: control words), trying the candidate classes
//                   from the smallest frame up: sequential per segment (a rebase depends on every step before it), one lane
//                   per segment;
//   k_plan_stripes  the stripe table of k_dp_tile; k_plan_coff  the row offsets of k_dp_generic; k_encode  ASCII -> base codes.
// The results are identical to the host planner's: tests compare them entry by entry (npr_batch_plan_check).
#include <hip/hip_runtime.h>

#include "npr_band.h"
#include "npr_device.h"
#include "npr_sched.h"

namespace npr {
namespace {

constexpr int BAND_THREADS = 256;

__global__ void __launch_bounds__(BAND_THREADS) k_plan_bands(PlanArgs a) {
    __shared__ long long s_cells[BAND_THREADS], s_gen[BAND_THREADS];
    __shared__ int s_w[BAND_THREADS], s_bad[BAND_THREADS], s_rough[BAND_THREADS];
    __shared__ int s_k[2];
    __shared__ int s_lo[BAND_THREADS + 1], s_hi[BAND_THREADS + 1];  // this tile's rows, [0] = the row before the tile
    for (int g = blockIdx.x; g < a.n_segs; g += gridDim.x) {
        const PlanSeg sg = a.segs[g];
        const PlanPoint *P = a.points + sg.point_first;
        const int D = sg.lX + sg.lY;
        long long cells = 0, gen = 0;
        int wmax = 0, bad = 0, rough = 0;
        for (int d0 = 0; d0 <= D; d0 += BAND_THREADS) {
            // the pieces this tile of anti-diagonals can fall into
            if (threadIdx.x < 2) s_k[threadIdx.x] = band_piece(P, sg.pieces, threadIdx.x == 0 ? d0 : min(d0 + BAND_THREADS - 1, D));
            __syncthreads();
            const int d = d0 + threadIdx.x;
            if (d <= D) {
                int lo = s_k[0], hi = s_k[1];
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (P[mid].d0() <= d) lo = mid; else hi = mid - 1;
                }
                const BandRow r = band_row_of_piece(a.fixed_mode, a.width, sg.lX, sg.lY, P[lo], P[lo + 1], d);
                a.lo[sg.band_off + d] = r.lo;
                a.n[sg.band_off + d] = r.n;
                cells += r.n > 0 ? r.n : 0;
                gen += r.n > 0 ? ((r.n + 3) & ~3) : 0;
                wmax = max(wmax, r.n);
                bad += r.n < 1;
                s_lo[threadIdx.x + 1] = r.lo, s_hi[threadIdx.x + 1] = r.lo + 2 * (r.n - 1);
            }
            __syncthreads();
            // both edges must move by exactly one cell per anti-diagonal (what the stripe table's binary searches and the
            // frame schedules rely on); a band that does not is flagged and takes the general paths
            if (d <= D && d > 0) {
                const int dl = s_lo[threadIdx.x + 1] - s_lo[threadIdx.x], dh = s_hi[threadIdx.x + 1] - s_hi[threadIdx.x];
                rough += (dl != 1 && dl != -1) || (dh != 1 && dh != -1);
            }
            __syncthreads();
            if (threadIdx.x == BAND_THREADS - 1) s_lo[0] = s_lo[BAND_THREADS], s_hi[0] = s_hi[BAND_THREADS];
            __syncthreads();
        }
        s_cells[threadIdx.x] = cells, s_gen[threadIdx.x] = gen, s_w[threadIdx.x] = wmax, s_bad[threadIdx.x] = bad, s_rough[threadIdx.x] = rough;
        __syncthreads();
        for (int k = BAND_THREADS / 2; k > 0; k >>= 1) {
            if (threadIdx.x < k) {
                s_cells[threadIdx.x] += s_cells[threadIdx.x + k], s_gen[threadIdx.x] += s_gen[threadIdx.x + k];
                s_w[threadIdx.x] = max(s_w[threadIdx.x], s_w[threadIdx.x + k]), s_bad[threadIdx.x] += s_bad[threadIdx.x + k];
                s_rough[threadIdx.x] += s_rough[threadIdx.x + k];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) a.summary[g] = SegSummary{s_cells[0], s_gen[0], s_w[0], s_bad[0], s_rough[0], 0};
        __syncthreads();
    }
}

// The schedule is a sequential scan per segment (every rebase depends on all steps before it), so the parallelism is ACROSS
// segments: a wavefront takes 16 of them and 16 of its lanes walk them in lockstep.  What a walking lane needs -- its own
// segment's band rows, one after the other -- is the worst pattern for memory, so the rows go through LDS: for every tile
// of 31 anti-diagonals all 64 lanes load the 32 rows (31 + the one looked ahead) of two segments per instruction, eight
// independent instructions per tile; the finished control words go back the same way, coalesced.  Few walking lanes per
// wavefront on purpose: the walk is a chain of dependent integer operations, what hides its latency is the number of
// wavefronts, not their width.  (One lane per segment straight from memory: 47 ms for the 12288 reads of the default bench;
// one wavefront per segment on the scalar unit: 40 ms -- a CU issues one scalar instruction per cycle whatever its
// occupancy; 64 segments per wavefront: 28 ms.)
#ifndef NPR_SCHED_SEGS
#define NPR_SCHED_SEGS 8  // segments per wavefront: 16 left the chip with less than one wavefront per SIMD on 12 k segments (46 -> 42 ms staging)
#endif
constexpr int SCHED_TILE = 31, SCHED_SEGS = NPR_SCHED_SEGS;
__global__ void __launch_bounds__(64) k_plan_sched(SchedArgs a) {
    __shared__ int t_lo[SCHED_SEGS][SCHED_TILE + 2], t_n[SCHED_SEGS][SCHED_TILE + 2];  // (row stride 33: conflict-free walks)
    __shared__ uint32_t t_w0[SCHED_SEGS][SCHED_TILE + 2], t_w1[SCHED_SEGS][SCHED_TILE + 2];
    const int lane = threadIdx.x;
    const int half = lane >> 5, row = lane & 31;
    for (int g0 = blockIdx.x * SCHED_SEGS; g0 < a.n_segs; g0 += gridDim.x * SCHED_SEGS) {
        const int g = g0 + lane;
        const bool have = lane < SCHED_SEGS && g < a.n_segs;
        PlanSeg sg{};
        if (have) sg = a.segs[g];
        const int64_t ctl_off = have ? a.ctl_off[g] : -1;
        const uint32_t cand = (have && ctl_off >= 0 && a.summary[g].bad == 0) ? a.cand[g] : 0u;
        const int max_width = have ? a.summary[g].max_width : 0;
        const int D = sg.lX + sg.lY;
        int cls = -1;
        uint32_t cells = 0;
        for (int c = 0; c < kSchedClasses; ++c) {
            const bool active = cls < 0 && ((cand >> c) & 1u);
            if (!__any(active)) continue;
            const int R = kSchedR[c], NW = kSchedNW[c], rshift = stair_rshift(R), C = 64 * R * NW;
            StairState st{0, 0, 0, 0, 0, 0};
            int ok = 0;
            if (active) ok = stair_begin(st, a.lo[sg.band_off], a.n[sg.band_off], max_width, R, NW) ? 1 : 0;
            int Dmax = active ? D : -1;
#pragma unroll
            for (int k = 32; k > 0; k >>= 1) Dmax = max(Dmax, __shfl_xor(Dmax, k, 64));
            for (int base = 0; base <= Dmax; base += SCHED_TILE) {
                if (!__any(ok && base <= D)) break;
                // rows base .. base + 31 of the walking segments into LDS: two segments per instruction
#pragma unroll
                for (int q = 0; q < SCHED_SEGS / 2; ++q) {
                    const int s2 = 2 * q + half;
                    const int walking = __shfl(ok && base <= D, s2, 64);
                    const int64_t off = __shfl(sg.band_off, s2, 64);
                    const int Ds = __shfl(D, s2, 64);
                    if (walking && base + row <= Ds) {
                        t_lo[s2][row] = a.lo[off + base + row];
                        t_n[s2][row] = a.n[off + base + row];
                    }
                }
                __syncthreads();
                const bool walk = ok && base <= D;
                if (walk) {
                    const int cnt = min(SCHED_TILE, D + 1 - base);
                    int lo_c = t_lo[lane][0], n_c = t_n[lane][0];  // the row in hand stays in registers: two LDS reads per step, not four
                    for (int i = 0; i < cnt && ok; ++i) {
                        const int lo_x = t_lo[lane][i + 1], n_x = t_n[lane][i + 1];
                        uint32_t w0 = 0, w1 = 0;
                        ok = stair_step(st, base + i, D, lo_c, n_c, lo_x, n_x, rshift, C, w0, w1) ? 1 : 0;
                        t_w0[lane][i] = w0, t_w1[lane][i] = w1;
                        lo_c = lo_x, n_c = n_x;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < SCHED_SEGS / 2; ++q) {
                    const int s2 = 2 * q + half;
                    const int done = __shfl(walk && ok, s2, 64);  // (a segment that failed in this tile is retried in the next class)
                    const int64_t off = __shfl(ctl_off, s2, 64);
                    const int Ds = __shfl(D, s2, 64);
                    if (done && row < SCHED_TILE && base + row <= Ds) {
                        uint2 *dst = reinterpret_cast<uint2 *>(a.ctl + 2 * (off + base + row));
                        *dst = make_uint2(t_w0[s2][row], t_w1[s2][row]);
                    }
                }
                __syncthreads();
            }
            if (active && ok) cls = c, cells = st.off;
        }
        if (have) a.cls[g] = cls, a.cells[g] = cls >= 0 ? static_cast<int64_t>(cells) : 0;
    }
}

// One wavefront per segment, a lane per stripe.  In a band whose edges move by one cell per anti-diagonal the first / last
// lattice column of the band never decreases with d, so the first anti-diagonal that reaches a stripe and the last one that
// still touches it are binary searches (a lane per segment walking all anti-diagonals and updating the table in memory took
// 62 ms for 9380 reads in the reference's band).  Bands flagged by k_plan_bands take the general walk.
__global__ void __launch_bounds__(64) k_plan_stripes(StripeArgs a) {
    const int lane = threadIdx.x;
    for (int i = blockIdx.x; i < a.count; i += gridDim.x) {
        const int g = a.seg_index[i];
        const PlanSeg sg = a.segs[g];
        const int32_t *lo = a.lo + sg.band_off, *n = a.n + sg.band_off;
        Stripe *out = a.stripes + a.tile_off[i];
        const int D = sg.lX + sg.lY, K = 64 * a.R, S = sg.lX / K + 1;
        if (a.summary[g].bad || a.summary[g].rough) {
            if (lane == 0) {
                a.rows[i] = stripe_ranges(lo, n, D, sg.lX, a.R, &out[1].df, &out[1].dl, static_cast<int>(sizeof(Stripe) / sizeof(int32_t)));
                stripe_fill(out, sg.lX, a.R);
            }
            continue;
        }
        int64_t rows = 0;
        for (int k0 = 0; k0 < S; k0 += 64) {
            const int k = k0 + lane;
            int df = 1, dl = 0;
            if (k < S) {
                const int X0 = k * K, X1 = X0 + K - 1;
                int a0 = 0, b0 = D;  // first d whose last band column, (d + lo) / 2 + n - 1, reaches X0
                while (a0 < b0) {
                    const int mid = (a0 + b0) >> 1;
                    if (((mid + lo[mid]) >> 1) + n[mid] - 1 >= X0) b0 = mid; else a0 = mid + 1;
                }
                df = a0;
                int a1 = 0, b1 = D;  // last d whose first band column is still <= X1
                while (a1 < b1) {
                    const int mid = (a1 + b1 + 1) >> 1;
                    if (((mid + lo[mid]) >> 1) <= X1) a1 = mid; else b1 = mid - 1;
                }
                dl = a1;
            }
            const int len = (k < S && dl >= df) ? dl - df + 1 : 0;
            int incl = len;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d, 64);
                if (lane >= d) incl += o;
            }
            if (k < S) {
                Stripe st;
                st.X = k * K, st.K = K, st.df = df, st.dl = dl;
                st.row0 = static_cast<uint32_t>(rows + incl - len);
                st.pad[0] = st.pad[1] = st.pad[2] = 0;
                out[1 + k] = st;
            }
            rows += __shfl(incl, 63, 64);
        }
        if (lane == 0) {
            Stripe hd{};
            hd.X = S, hd.K = static_cast<int32_t>(rows);
            out[0] = hd;
            a.rows[i] = rows;
        }
    }
}

__global__ void __launch_bounds__(64) k_plan_coff(CoffArgs a) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_segs) return;
    const PlanSeg sg = a.segs[g];
    const int32_t *n = a.n + sg.band_off;
    uint32_t *co = a.coff + sg.band_off;
    uint64_t off = 0;
    for (int d = 0; d <= sg.lX + sg.lY; ++d) {
        co[d] = static_cast<uint32_t>(off);
        off += (static_cast<uint64_t>(n[d] > 0 ? n[d] : 0) + 3) & ~uint64_t(3);  // 16-byte aligned rows
    }
}

// The packed lane masks of every row of every stripe of the k_dp_tile tasks (npr_sched.h tile_row_word): one workgroup
// per task, its wavefronts take the stripes round-robin, a lane per row.
__global__ void __launch_bounds__(256) k_plan_rowmask(RowMaskArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = blockIdx.x; i < a.count; i += gridDim.x) {
        const PlanSeg sg = a.segs[a.seg_index[i]];
        const int32_t *lo = a.lo + sg.band_off, *n = a.n + sg.band_off;
        const Stripe *tab = a.stripes + a.tile_off[i];
        uint32_t *out = a.out + a.mask_off[i];
        const int S = tab[0].X;
        for (int s = wv; s < S; s += 4) {
            const Stripe st = tab[1 + s];
            for (int d = st.df + lane; d <= st.dl; d += 64) out[st.row0 + static_cast<uint32_t>(d - st.df)] = tile_row_word(d, lo[d], n[d], st.X);
        }
    }
}

__device__ __forceinline__ uint32_t code_of(uint32_t c) {
    c &= 0xdfu;  // upper case
    return c == 'A' ? 0u : (c == 'C' ? 1u : (c == 'G' ? 2u : (c == 'T' ? 3u : 4u)));
}
__device__ __forceinline__ uint32_t code4(uint32_t v) {
    return code_of(v & 0xffu) | code_of((v >> 8) & 0xffu) << 8 | code_of((v >> 16) & 0xffu) << 16 | code_of(v >> 24) << 24;
}
// ASCII -> base codes 0..4 in place, 16 bases per thread and step (the buffer is 16-byte aligned)
__global__ void __launch_bounds__(256) k_encode(uint8_t *seq, int64_t n) {
    const int64_t words = n / 16;
    uint4 *w = reinterpret_cast<uint4 *>(seq);
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < words; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        uint4 v = w[i];
        v.x = code4(v.x), v.y = code4(v.y), v.z = code4(v.z), v.w = code4(v.w);
        w[i] = v;
    }
    if (blockIdx.x == 0)
        for (int64_t i = words * 16 + threadIdx.x; i < n; i += blockDim.x) seq[i] = static_cast<uint8_t>(code_of(seq[i]));
}

}  // namespace

int launch_plan_bands(const PlanArgs &a, void *stream) {
    const int grid = a.n_segs < 65536 ? (a.n_segs > 0 ? a.n_segs : 1) : 65536;
    hipLaunchKernelGGL(k_plan_bands, dim3(grid), dim3(BAND_THREADS), 0, static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}
int launch_plan_sched(const SchedArgs &a, void *stream) {
    const int grid = (a.n_segs + SCHED_SEGS - 1) / SCHED_SEGS > 0 ? (a.n_segs + SCHED_SEGS - 1) / SCHED_SEGS : 1;
    hipLaunchKernelGGL(k_plan_sched, dim3(grid), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}
int launch_plan_stripes(const StripeArgs &a, void *stream) {
    const int grid = a.count < 65536 ? (a.count > 0 ? a.count : 1) : 65536;
    hipLaunchKernelGGL(k_plan_stripes, dim3(grid), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}
int launch_plan_rowmask(const RowMaskArgs &a, void *stream) {
    if (a.count <= 0) return 0;
    hipLaunchKernelGGL(k_plan_rowmask, dim3(a.count < 65536 ? a.count : 65536), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}
int launch_plan_coff(const CoffArgs &a, void *stream) {
    hipLaunchKernelGGL(k_plan_coff, dim3((a.n_segs + 63) / 64 > 0 ? (a.n_segs + 63) / 64 : 1), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}
int launch_encode(uint8_t *seq, int64_t n, void *stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_encode, dim3(4096), dim3(256), 0, static_cast<hipStream_t>(stream), seq, n);
    return static_cast<int>(hipGetLastError());
}

}  // namespace npr
