// npr_stats.hip -- k_align_stats: the per-read integer reductions of the reference's post-alignment analyses, on the
// device, from alignments that already lie there.
//
// nanopore/analyses/coverage.py:10-95 (ReadAlignmentCoverageCounter: matches, mismatches, aligned pairs against N, number
// and total length of read insertions / deletions, with the leading / trailing indels a global alignment adds),
// substitutions.py:9-56 (5 x 5 substitution counts over aligned pairs) and indels.py:9-45 (insertion / deletion counts)
// walk every aligned pair of every SAM record in Python.  All of it is integer work over (cigar, reference bases, read
// bases): one wavefront per read walks the run-length cigar 64 ops at a time -- lanes scan the op extents into start
// coordinates, the M columns of the chunk are dealt out evenly over the lanes (a binary search in the chunk's prefix sums),
// each lane compares its column's two bases, ballots count matches / mismatches, 25 LDS counters take the substitution
// matrix -- while the gaps between aligned pairs are classified in a scalar loop over the chunk's ops.
// Input is what npr_batch_finish leaves on the device (packed ops from k_mea_gather, base codes of the tasks' segments)
// or, for any other SAM file, the same uploaded by npr_align_stats.  Counts are exact integers: the tests compare them
// with an independent CPU counter.
#include <hip/hip_runtime.h>

#include "npr_device.h"

namespace npr {
namespace {

constexpr int WAVE = 64;

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int &total) {
    int s = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const int o = __shfl_up(s, d, WAVE);
        if (lane >= d) s += o;
    }
    total = __shfl(s, WAVE - 1, WAVE);
    return s - v;
}

__global__ void __launch_bounds__(WAVE) k_align_stats(StatsArgs a) {
    __shared__ int sx[WAVE], sy[WAVE], sm[WAVE + 1];
    __shared__ int sub[25];
    const int lane = threadIdx.x;
    for (int r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
        const int64_t o0 = a.ops_off[r];
        const int nops = static_cast<int>(a.ops_off[r + 1] - o0);
        const int nseg = a.seg_off[r + 1] - a.seg_off[r];
        if (lane < 25) sub[lane] = 0;
        const StatsSeg *segs = a.segs + a.seg_off[r];
        __syncthreads();
        int x = 0, y = 0;                    // window coordinates of the next op
        int matches = 0, mismatches = 0, ns = 0, pairs = 0;
        int seen_m = 0, pend_i = 0, pend_d = 0;  // gap since the last aligned pair
        int n_ins = 0, ins_len = 0, n_del = 0, del_len = 0, lead_i = 0, lead_d = 0;
        for (int base = 0; base < nops; base += WAVE) {
            const int cnt = min(WAVE, nops - base);
            const uint32_t w = lane < cnt ? a.ops[o0 + base + lane] : 0u;
            const int op = static_cast<int>(w & 3u), len = static_cast<int>(w >> 2);
            int tx, ty, tm;
            const int ex = wave_excl_scan(op != NPR_OP_I ? len : 0, lane, tx);
            const int ey = wave_excl_scan(op != NPR_OP_D ? len : 0, lane, ty);
            const int em = wave_excl_scan(op == NPR_OP_M ? len : 0, lane, tm);
            sx[lane] = x + ex, sy[lane] = y + ey, sm[lane] = em;
            if (lane == 0) sm[WAVE] = tm;
            // gaps between aligned pairs: scalar walk over the chunk's ops
            for (int i = 0; i < cnt; ++i) {
                const uint32_t wi = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w), i));
                const int oi = static_cast<int>(wi & 3u), li = static_cast<int>(wi >> 2);
                if (li == 0) continue;
                if (oi == NPR_OP_M) {
                    if (seen_m) {
                        n_ins += pend_i > 0, ins_len += pend_i, n_del += pend_d > 0, del_len += pend_d;
                    } else {
                        lead_i = pend_i, lead_d = pend_d;
                    }
                    seen_m = 1, pend_i = 0, pend_d = 0;
                } else if (oi == NPR_OP_I) {
                    pend_i += li;
                } else {
                    pend_d += li;
                }
            }
            __syncthreads();
            // the chunk's M columns, dealt out over the lanes
            for (int t = 0; t < tm; t += WAVE) {
                const int c = t + lane;
                int hit = 0, rb = 4, qb = 4;
                if (c < tm) {
                    int lo = 0, hi = cnt - 1;  // last op whose exclusive M prefix is <= c
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (sm[mid] <= c) lo = mid; else hi = mid - 1;
                    }
                    const int xx = sx[lo] + (c - sm[lo]), yy = sy[lo] + (c - sm[lo]);
                    for (int s = 0; s < nseg; ++s) {
                        const StatsSeg &g = segs[s];
                        if (xx >= g.xs && xx < g.xe && yy >= g.ys && yy < g.ye) {
                            rb = a.seq[g.x_off + (xx - g.xs)];
                            qb = a.seq[g.y_off + (yy - g.ys)];
                            hit = 1;
                            break;
                        }
                    }
                    if (!hit) rb = qb = 4;  // an aligned pair outside every uploaded piece counts against N
                    atomicAdd(&sub[min(rb, 4) * 5 + min(qb, 4)], 1);
                }
                const bool in = c < tm;
                matches += __popcll(__ballot(in && rb < 4 && rb == qb));
                mismatches += __popcll(__ballot(in && rb < 4 && qb < 4 && rb != qb));
                pairs += __popcll(__ballot(in));
            }
            x += tx, y += ty;
            __syncthreads();
        }
        ns = pairs - matches - mismatches;
        int32_t *out = a.out + static_cast<int64_t>(r) * NPR_STATS_WORDS;
        if (lane == 0) {
            out[0] = matches, out[1] = mismatches, out[2] = ns, out[3] = pairs;
            out[4] = n_ins, out[5] = ins_len, out[6] = n_del, out[7] = del_len;
            out[8] = seen_m ? lead_i : pend_i, out[9] = seen_m ? lead_d : pend_d;   // before the first aligned pair
            out[10] = seen_m ? pend_i : 0, out[11] = seen_m ? pend_d : 0;           // after the last one
            out[12] = x, out[13] = y;                                               // reference / read bases the cigar consumes
            out[14] = 0;
        }
        if (lane < 25) out[15 + lane] = sub[lane];
        __syncthreads();
    }
}

// Expected base counts per reference position (nanopore/analyses/marginAlignSnpCaller.py:150-155): every
// posterior pair (x, y, p) of a selected read adds p to the count of the read's base y at reference position x.  The pairs
// are where the DP kernels left them; a workgroup takes a task, its threads the task's pairs; 64-bit integer atomics into a
// table of 4 counts per reference position in fixed point (p * 2^40: exact for an fp32 p >= 2^-17, so a sum is the exact sum
// of its terms whatever order the atomics land in -- the same bits from run to run, which fp64 atomics did not give),
// plus a byte that says the position was seen at all.
__global__ void __launch_bounds__(256) k_base_expectations(ExpectArgs a) {
    for (int t = blockIdx.x; t < a.ntasks; t += gridDim.x) {
        const Task &tk = a.tasks[t];
        const int r = tk.read;
        if (a.use && !a.use[r]) continue;
        const int n = min(a.outs[t].npairs, tk.pair_cap);
        const int64_t target = a.target[r];  // table row of the window's first reference position
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int x = a.px[tk.pair_off + i], y = a.py[tk.pair_off + i];
            const int code = a.seq[tk.y_off + (y - tk.ys)];
            a.seen[target + x] = 1;
            if (code < 4) atomicAdd(a.expect + 4 * (target + x) + code, __float2ull_rz(a.pp[tk.pair_off + i] * EXPECT_FIXED_ONE));
        }
    }
}

// NPR_MODE_RESCORE_ORIGINAL (npr_device.h RescoreArgs): the guide's M runs as a table over the reference positions of each read's window ...
__global__ void __launch_bounds__(256) k_rescore_table(RescoreArgs a) {
    for (int r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
        int32_t *gy = a.gy + a.gx_off[r];
        const int64_t span = a.gx_off[r + 1] - a.gx_off[r];
        // short runs (a noisy read's: a few dozen columns): a thread each; long ones (a near-exact guide, = / X collapsed into M: tens of thousands of
        // columns) by the whole workgroup, so that no single lane writes a run alone while npr_batch_create holds the stream
        constexpr int LONG_RUN = 64;
        for (int64_t q = a.run_off[r] + threadIdx.x; q < a.run_off[r + 1]; q += blockDim.x) {
            const int x0 = a.runs[3 * q], y0 = a.runs[3 * q + 1], len = a.runs[3 * q + 2];
            if (len > LONG_RUN) continue;
            for (int t = 0; t < len; ++t)
                if (x0 + t >= 0 && x0 + t < span) gy[x0 + t] = y0 + t;
        }
        for (int64_t q = a.run_off[r]; q < a.run_off[r + 1]; ++q) {
            const int len = a.runs[3 * q + 2];
            if (len <= LONG_RUN) continue;
            const int x0 = a.runs[3 * q], y0 = a.runs[3 * q + 1];
            for (int t = threadIdx.x; t < len; t += blockDim.x)
                if (x0 + t >= 0 && x0 + t < span) gy[x0 + t] = y0 + t;
        }
    }
}
// ... and the posterior mass of the pairs that lie on it, per read, in fixed point (one 64-bit atomic per wavefront and task)
__global__ void __launch_bounds__(256) k_rescore_sum(RescoreArgs a) {
    const float one = __builtin_ldexpf(1.0f, a.shift);
    for (int t = blockIdx.x; t < a.ntasks; t += gridDim.x) {
        const Task &tk = a.tasks[t];
        const int r = tk.read;
        const int n = min(a.outs[t].npairs, tk.pair_cap);
        const int32_t *gy = a.gy + a.gx_off[r];
        const int64_t span = a.gx_off[r + 1] - a.gx_off[r];
        unsigned long long local = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int x = a.px[tk.pair_off + i];
            if (x >= 0 && x < span && gy[x] == a.py[tk.pair_off + i]) local += __float2ull_rz(a.pp[tk.pair_off + i] * one);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned lo = __shfl_xor(static_cast<unsigned>(local), o, WAVE), hi = __shfl_xor(static_cast<unsigned>(local >> 32), o, WAVE);
            local += (static_cast<unsigned long long>(hi) << 32) | lo;
        }
        if ((threadIdx.x & (WAVE - 1)) == 0 && local) atomicAdd(a.sum + r, local);
    }
}

}  // namespace

int launch_rescore_table(const RescoreArgs &a, void *stream) {
    const int g1 = a.n_reads < 16384 ? (a.n_reads > 0 ? a.n_reads : 1) : 16384;
    hipLaunchKernelGGL(k_rescore_table, dim3(g1), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}
int launch_rescore_sum(const RescoreArgs &a, void *stream) {
    const int g2 = a.ntasks < 8192 ? (a.ntasks > 0 ? a.ntasks : 1) : 8192;
    hipLaunchKernelGGL(k_rescore_sum, dim3(g2), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}

int launch_base_expectations(const ExpectArgs &a, void *stream) {
    const int grid = a.ntasks < 8192 ? (a.ntasks > 0 ? a.ntasks : 1) : 8192;
    hipLaunchKernelGGL(k_base_expectations, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}

int launch_align_stats(const StatsArgs &a, void *stream) {
    const int grid = a.n_reads < 16384 ? (a.n_reads > 0 ? a.n_reads : 1) : 16384;
    hipLaunchKernelGGL(k_align_stats, dim3(grid), dim3(WAVE), 0, static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}

}  // namespace npr
