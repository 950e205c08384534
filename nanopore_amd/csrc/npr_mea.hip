// npr_mea.hip -- the maximum-expected-accuracy chain and its cigar on the device (the
// reference gets them from cactus_realign, utils.py:587-605).  Integer arithmetic throughout, so the result is the
// one npr_host.cpp's mea_cigar() produces from the same posterior pairs, bit for bit; what changes is where it runs:
// the pairs (about 12 bytes per reference base and read) stay in HBM and only the run-length encoded ops cross PCIe.
//
//   k_mea_count    per pair: quantise the posterior (floor(p * 1e7)), count the pair on its reference position,
//                  add the quantum to its read position's column sum
//   k_mea_scan     per read: exclusive scan of the counts -> first sorted slot of every reference position
//   k_mea_scatter  per pair: move to its reference position's group (order inside a group arbitrary)
//   k_mea_chain    one wavefront per read: groups in reference order; inside a 64-pair chunk the lanes sort their
//                  groups by read position, weigh the pairs (posterior - gapGamma * gap mass of row and column) and
//                  drop those not above matchGamma; the heaviest chain ending below every read position y lives in
//                  an LDS ring keyed by y (a monotone prefix maximum: a query is one read, an insert overwrites the
//                  run of entries the new chain beats, all lanes at once); ties go to the pair that sorts last
//   k_mea_trace    one lane per read: walk the back pointers from the best chain's last pair, writing the ops
//                  backwards (run-length merged) into the read's scratch, and sum the chain's posterior mass
//   k_mea_gather   dense copy of every read's ops for one D2H
#include <hip/hip_runtime.h>

#include "npr_device.h"

namespace npr {
namespace {

constexpr int64_t P1 = PROB_ONE;
constexpr int WAVE = 64;

__device__ __forceinline__ int rdlane(int v, int j) { return __builtin_amdgcn_readlane(v, j); }
__device__ __forceinline__ int64_t rdlane64(int64_t v, int j) {
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(v), j);
    const int hi = __builtin_amdgcn_readlane(static_cast<int>(v >> 32), j);
    return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}
__device__ __forceinline__ bool beats(int64_t s, int w, int64_t os, int ow) { return s > os || (s == os && w > ow); }

__global__ void __launch_bounds__(256) k_mea_count(MeaArgs a) {
    for (int t = blockIdx.x; t < a.ntasks; t += gridDim.x) {
        const Task &tk = a.tasks[t];
        const int n = min(a.outs[t].npairs, tk.pair_cap);
        const int r = tk.read;
        const int64_t rx = a.rx_off[r], ry = a.ry_off[r];
        const int lX = static_cast<int>(a.rx_off[r + 1] - rx) - 1, lY = static_cast<int>(a.ry_off[r + 1] - ry);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int x = a.px[tk.pair_off + i], y = a.py[tk.pair_off + i];
            if (x < 0 || x >= lX || y < 0 || y >= lY) {
                a.read_flag[r] = NPR_ERR_INVALID;
                continue;
            }
            const int q = static_cast<int>(floor(static_cast<double>(a.pp[tk.pair_off + i]) * static_cast<double>(P1)));
            atomicAdd(a.cnt + rx + x, 1);
            atomicAdd(a.colsum + ry + y, q);
        }
    }
}

// start[x] = number of pairs of the read on reference positions < x, for x in [0, lX]
__global__ void __launch_bounds__(256) k_mea_scan(MeaArgs a) {
    __shared__ int part[256];
    const int r = blockIdx.x;
    const int64_t rx = a.rx_off[r];
    const int n = static_cast<int>(a.rx_off[r + 1] - rx);  // lX + 1 entries (the last count is zero)
    const int per = (n + 255) / 256, lo = min(n, static_cast<int>(threadIdx.x) * per), hi = min(n, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += a.cnt[rx + i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int v = static_cast<int>(threadIdx.x) >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int i = lo; i < hi; ++i) {
        a.start[rx + i] = run;
        run += a.cnt[rx + i];
    }
}

__global__ void __launch_bounds__(256) k_mea_scatter(MeaArgs a) {
    for (int t = blockIdx.x; t < a.ntasks; t += gridDim.x) {
        const Task &tk = a.tasks[t];
        const int n = min(a.outs[t].npairs, tk.pair_cap);
        const int r = tk.read;
        const int64_t rx = a.rx_off[r], rp = a.rp_off[r];
        const int lX = static_cast<int>(a.rx_off[r + 1] - rx) - 1, lY = static_cast<int>(a.ry_off[r + 1] - a.ry_off[r]);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int x = a.px[tk.pair_off + i], y = a.py[tk.pair_off + i];
            if (x < 0 || x >= lX || y < 0 || y >= lY) continue;
            const int q = static_cast<int>(floor(static_cast<double>(a.pp[tk.pair_off + i]) * static_cast<double>(P1)));
            const int64_t pos = rp + a.start[rx + x] + (atomicSub(a.cnt + rx + x, 1) - 1);
            a.sx[pos] = x, a.sy[pos] = y, a.sq[pos] = q;
        }
    }
}

// LDS: ring of `ring` (score int64, who int32) entries keyed by read position, then the 64-lane permutation buffers
__global__ void __launch_bounds__(WAVE) k_mea_chain(MeaArgs a) {
    extern __shared__ __align__(16) char lds[];
    const int RING = a.ring, MASK = RING - 1;
    int64_t *const rs = reinterpret_cast<int64_t *>(lds);
    int *const rw = reinterpret_cast<int *>(rs + RING);
    int64_t *const tw = reinterpret_cast<int64_t *>(rw + RING);  // weights in sorted lane order
    int *const ty = reinterpret_cast<int *>(tw + WAVE);
    int *const tq = ty + WAVE;
    int *const tk = tq + WAVE;
    const int lane = threadIdx.x;
    const int r = blockIdx.x;
    const int64_t rp = a.rp_off[r], ry = a.ry_off[r];
    const int n = a.read_flag[r] ? 0 : static_cast<int>(a.rp_off[r + 1] - rp);  // (flagged: the scatter was incomplete)
    int *const sx = a.sx + rp, *const sy = a.sy + rp, *const sq = a.sq + rp, *const back = a.back + rp;
    const int64_t floor_w = static_cast<int64_t>(floor(a.match_gamma * static_cast<double>(P1)));
    int ytop = -1;  // largest key the ring holds; beyond it the prefix maximum is `top`
    int64_t top_s = 0;
    int top_w = -1;
    int flag = 0;
    for (int base = 0; base < n && !flag;) {
        const int pos = base + lane;
        const bool in = pos < n;
        const int x = in ? sx[pos] : -1, xn = pos + 1 < n ? sx[pos + 1] : -2;
        int y = in ? sy[pos] : 0, q = in ? sq[pos] : 0;
        const uint64_t ends = __ballot(in && x != xn);  // last lane of every reference position's group
        if (!ends) {
            flag = NPR_ERR_CAPACITY;  // more than 64 pairs on one reference position: cannot happen above a 0.01 threshold
            break;
        }
        const int valid = 64 - __builtin_clzll(ends);
        const bool act = lane < valid;
        // group statistics: rank by read position, first lane of the group, posterior mass of the row
        int rank = 0, before = 0, rowsum = 0;
        for (int j = 0; j < valid; ++j) {
            const int xj = rdlane(x, j), yj = rdlane(y, j), qj = rdlane(q, j);
            const bool same = xj == x;
            rank += (same && yj < y) ? 1 : 0;
            before += (same && j < lane) ? 1 : 0;
            rowsum += same ? qj : 0;
        }
        const int colsum = act ? a.colsum[ry + y] : 0;
        const int64_t gap = max(P1 - rowsum, int64_t(0)) + max(P1 - colsum, int64_t(0));
        const int64_t w0 = q - static_cast<int64_t>(floor(a.gap_gamma * static_cast<double>(gap)));
        __syncthreads();
        if (act) {
            const int to = lane - before + rank;  // sorted place inside the chunk
            tw[to] = w0, ty[to] = y, tq[to] = q, tk[to] = w0 > floor_w ? 1 : 0;
        }
        __syncthreads();
        const int64_t w = tw[lane];
        y = ty[lane], q = tq[lane];
        const bool keep = act && tk[lane] != 0;
        if (act) sy[pos] = y, sq[pos] = q;  // (x, y)-sorted from here on: the trace reads them back
        uint64_t gm = ends & (valid == 64 ? ~0ull : ((1ull << valid) - 1));
        int gs = 0;
        while (gm) {
            const int ge = __builtin_ctzll(gm);
            gm &= gm - 1;
            const bool mine = keep && lane >= gs && lane <= ge;
            // heaviest chain over pairs already inserted with read position < y
            int64_t bs = 0;
            int bw = -1;
            const int key = y - 1;
            if (mine && key >= 0) {
                if (key > ytop) {
                    bs = top_s, bw = top_w;
                } else if (key <= ytop - RING) {
                    flag = NPR_ERR_CAPACITY;
                } else {
                    bs = rs[key & MASK], bw = rw[key & MASK];
                }
            }
            const int64_t total = w + bs;
            if (mine) back[pos] = bw;
            __syncthreads();
            uint64_t km = __ballot(mine);
            while (km) {
                const int i = __builtin_ctzll(km);
                km &= km - 1;
                const int vy = rdlane(y, i), vw = base + i;
                const int64_t vs = rdlane64(total, i);
                if (vy > ytop) {
                    const int fill = min(vy - ytop - 1, RING);
                    for (int k = vy - fill + lane; k < vy; k += WAVE) rs[k & MASK] = top_s, rw[k & MASK] = top_w;
                    if (beats(vs, vw, top_s, top_w)) top_s = vs, top_w = vw;
                    if (lane == 0) rs[vy & MASK] = top_s, rw[vy & MASK] = top_w;
                    ytop = vy;
                } else {
                    if (vy <= ytop - RING) flag = NPR_ERR_CAPACITY;
                    for (int k0 = vy; k0 <= ytop; k0 += WAVE) {  // the entries this chain beats are a run starting at vy
                        const int k = k0 + lane;
                        const bool inr = k <= ytop;
                        const bool bt = inr && beats(vs, vw, rs[k & MASK], rw[k & MASK]);
                        if (bt) rs[k & MASK] = vs, rw[k & MASK] = vw;
                        if (__ballot(bt) != __ballot(inr)) break;
                    }
                    if (beats(vs, vw, top_s, top_w)) top_s = vs, top_w = vw;
                }
                __syncthreads();
            }
            gs = ge + 1;
        }
        flag = __any(flag) ? NPR_ERR_CAPACITY : 0;
        base += valid;
    }
    if (lane == 0) {
        a.best_who[r] = top_w;
        if (flag && a.read_flag[r] == 0) a.read_flag[r] = flag;
    }
}

__global__ void __launch_bounds__(WAVE) k_mea_trace(MeaArgs a) {
    const int r = blockIdx.x * WAVE + threadIdx.x;
    if (r >= a.n_reads) return;
    const int64_t rp = a.rp_off[r];
    const int *sx = a.sx + rp, *sy = a.sy + rp, *sq = a.sq + rp, *back = a.back + rp;
    const int lX = static_cast<int>(a.rx_off[r + 1] - a.rx_off[r]) - 1, lY = static_cast<int>(a.ry_off[r + 1] - a.ry_off[r]);
    int2 *const lo = reinterpret_cast<int2 *>(a.ops_tmp) + a.ot_off[r];
    int2 *p = reinterpret_cast<int2 *>(a.ops_tmp) + a.ot_off[r + 1];  // filled from the end
    int hop = -1, hlen = 0;
    auto emit = [&](int op, int len) {
        if (len <= 0) return;
        if (op == hop) {
            hlen += len;
            return;
        }
        if (hop >= 0 && p > lo) *--p = make_int2(hop, hlen);
        hop = op, hlen = len;
    };
    int cx = lX, cy = lY, len = 0;
    int64_t mass = 0;
    if (a.read_flag[r] == 0)
        for (int i = a.best_who[r]; i >= 0; i = back[i]) {
            const int x = sx[i], y = sy[i];
            emit(NPR_OP_I, cy - y - 1);  // backwards: the pair's M comes last in its (D, I, M) triple
            emit(NPR_OP_D, cx - x - 1);
            emit(NPR_OP_M, 1);
            cx = x, cy = y, mass += sq[i], ++len;
        }
    emit(NPR_OP_I, cy);
    emit(NPR_OP_D, cx);
    if (hop >= 0 && p > lo) *--p = make_int2(hop, hlen);
    a.n_ops[r] = static_cast<int>(reinterpret_cast<int2 *>(a.ops_tmp) + a.ot_off[r + 1] - p);
    a.chain_len[r] = len;
    a.chain_mass[r] = mass;
}

__global__ void __launch_bounds__(256) k_mea_gather(MeaArgs a) {
    for (int r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
        const int n = static_cast<int>(a.od_off[r + 1] - a.od_off[r]);
        const int2 *src = reinterpret_cast<const int2 *>(a.ops_tmp) + a.ot_off[r + 1] - a.n_ops[r];
        int2 *dst = reinterpret_cast<int2 *>(a.ops_dense) + a.od_off[r];
        for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    }
}

}  // namespace

size_t mea_chain_lds_bytes(int ring) { return static_cast<size_t>(ring) * 12 + WAVE * (8 + 4 + 4 + 4); }

int launch_mea_sort(const MeaArgs &a, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int tg = a.ntasks < 8192 ? (a.ntasks > 0 ? a.ntasks : 1) : 8192;
    hipLaunchKernelGGL(k_mea_count, dim3(tg), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_mea_scan, dim3(a.n_reads), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_mea_scatter, dim3(tg), dim3(256), 0, s, a);
    return static_cast<int>(hipGetLastError());
}

int launch_mea_chain(const MeaArgs &a, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = mea_chain_lds_bytes(a.ring);
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_mea_chain), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return static_cast<int>(e);
    hipLaunchKernelGGL(k_mea_chain, dim3(a.n_reads), dim3(WAVE), lds, s, a);
    hipLaunchKernelGGL(k_mea_trace, dim3((a.n_reads + WAVE - 1) / WAVE), dim3(WAVE), 0, s, a);
    return static_cast<int>(hipGetLastError());
}

int launch_mea_gather(const MeaArgs &a, void *stream) {
    const int g = a.n_reads < 8192 ? a.n_reads : 8192;
    hipLaunchKernelGGL(k_mea_gather, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}

}  // namespace npr
