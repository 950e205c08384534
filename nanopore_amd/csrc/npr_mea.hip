// npr_mea.hip -- the maximum-expected-accuracy chain and its cigar on the device (the
// reference gets them from cactus_realign, utils.py:587-605).  Integer arithmetic throughout, so the result is the
// one npr_host.cpp's mea_cigar() produces from the same posterior pairs, bit for bit; what changes is where it runs:
// the pairs (about 12 bytes per reference base and read) stay in HBM and only the run-length encoded ops cross PCIe.
//
//   k_mea_sort_lds count + scan + scatter below in one kernel, one workgroup per read with its tables in LDS; used
//                  when every read's span fits (the three kernels below otherwise: records chained over long spans)
//   k_mea_count    per pair: quantise the posterior (floor(p * 1e7)), count the pair on its reference position,
//                  add the quantum to its read position's column sum
//   k_mea_scan     per read: exclusive scan of the counts -> first sorted slot of every reference position
//   k_mea_scatter  per pair: move to its reference position's group (order inside a group arbitrary)
//   k_mea_chain_win  one wavefront per read: groups in reference order; inside a 64-pair chunk the lanes sort their
//                  groups by read position, weigh the pairs (posterior - gapGamma * gap mass of row and column) and
//                  drop those not above matchGamma; the heaviest chain ending below every read position y is a
//                  monotone prefix maximum kept in registers for a window of 128 read positions (a query is a
//                  readlane, an insert one compare-and-select); ties go to the pair that sorts last
//   k_mea_chain    the same with the prefix maximum in an LDS ring of any length, for the reads whose pairs reach
//                  back further than the window (none on usual data)
//   k_mea_trace    one wavefront per read: walk the back pointers from the best chain's last pair, writing the ops
//                  backwards (run-length merged) into the read's scratch, and sum the chain's posterior mass
//   k_mea_gather   dense copy of every read's ops, one word each, for one D2H
#include <hip/hip_runtime.h>

#include "npr_device.h"

namespace npr {
namespace {

constexpr int64_t P1 = PROB_ONE;
constexpr int WAVE = 64;
constexpr int SORT_THREADS = 1024;  // k_mea_sort_lds: its loops wait on memory, so many wavefronts per read

__device__ __forceinline__ int rdlane(int v, int j) { return __builtin_amdgcn_readlane(v, j); }
__device__ __forceinline__ int64_t rdlane64(int64_t v, int j) {
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(v), j);
    const int hi = __builtin_amdgcn_readlane(static_cast<int>(v >> 32), j);
    return (static_cast<int64_t>(hi) << 32) | static_cast<uint32_t>(lo);
}
// the kernel's workgroups are one wavefront: LDS operations of a wavefront complete in issue order, so cross-lane
// traffic through LDS needs the compiler to keep the order, not a hardware barrier
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
    __shared__ int wsum[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t rx = a.rx_off[r];
    const int n = static_cast<int>(a.rx_off[r + 1] - rx);  // lX + 1 entries (the last count is zero)
    int carry = 0;
    for (int base = 0; base < n; base += 256) {
        const int i = base + tid;
        const int v = i < n ? a.cnt[rx + i] : 0;
        int s = v;
        for (int o = 1; o < WAVE; o <<= 1) {
            const int t = __shfl_up(s, o);
            if (lane >= o) s += t;
        }
        if (lane == WAVE - 1) wsum[wv] = s;
        __syncthreads();
        int before = 0, total = 0;
        for (int k = 0; k < 4; ++k) before += k < wv ? wsum[k] : 0, total += wsum[k];
        if (i < n) a.start[rx + i] = carry + before + s - v;
        carry += total;
        __syncthreads();
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

// count + scan + scatter of one read in LDS (one workgroup per read, its tasks' pairs read three times, from L2 after
// the first): column sums first, then -- in the same LDS -- the per-position counts, their exclusive scan in place and
// the scatter with the scanned counts as fill pointers.  For reads whose longer span fits the LDS (4 bytes per base).
__global__ void __launch_bounds__(SORT_THREADS) k_mea_sort_lds(MeaArgs a) {
    extern __shared__ int h[];
    __shared__ int wsum[SORT_THREADS / WAVE];
    const int r = a.order[blockIdx.x], tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t rx = a.rx_off[r], ry = a.ry_off[r], rp = a.rp_off[r];
    const int lX = static_cast<int>(a.rx_off[r + 1] - rx) - 1, lY = static_cast<int>(a.ry_off[r + 1] - ry);
    if (a.rp_off[r + 1] == rp) return;
    const int ft = a.read_first[r], nt = a.read_ntasks[r];
    int bad = 0;
    // the pairs of every task of the read, f(x, y, p-quantum)
    auto for_pairs = [&](auto f) {
        for (int s = 0; s < nt; ++s) {
            const int t = a.task_of[ft + s];
            const Task &tk = a.tasks[t];
            const int n = min(a.outs[t].npairs, tk.pair_cap);
            for (int i = tid; i < n; i += SORT_THREADS) {
                const int x = a.px[tk.pair_off + i], y = a.py[tk.pair_off + i];
                if (x < 0 || x >= lX || y < 0 || y >= lY) {
                    bad = 1;
                    continue;
                }
                f(x, y, static_cast<int>(floor(static_cast<double>(a.pp[tk.pair_off + i]) * static_cast<double>(P1))));
            }
        }
    };
    for (int i = tid; i < lY; i += SORT_THREADS) h[i] = 0;
    __syncthreads();
    for_pairs([&](int, int y, int q) { atomicAdd(&h[y], q); });
    __syncthreads();
    for (int i = tid; i < lY; i += SORT_THREADS) a.colsum[ry + i] = h[i];
    __syncthreads();
    for (int i = tid; i <= lX; i += SORT_THREADS) h[i] = 0;
    __syncthreads();
    for_pairs([&](int x, int, int) { atomicAdd(&h[x], 1); });
    __syncthreads();
    int carry = 0;
    for (int base = 0; base <= lX; base += SORT_THREADS) {
        const int i = base + tid;
        const int v = i <= lX ? h[i] : 0;
        int sc = v;
        for (int o = 1; o < WAVE; o <<= 1) {
            const int t = __shfl_up(sc, o);
            if (lane >= o) sc += t;
        }
        if (lane == WAVE - 1) wsum[wv] = sc;
        __syncthreads();
        int before = 0, total = 0;
        for (int k = 0; k < SORT_THREADS / WAVE; ++k) before += k < wv ? wsum[k] : 0, total += wsum[k];
        if (i <= lX) h[i] = carry + before + sc - v;
        carry += total;
        __syncthreads();
    }
    for_pairs([&](int x, int y, int q) {
        const int64_t pos = rp + atomicAdd(&h[x], 1);
        a.sx[pos] = x, a.sy[pos] = y, a.sq[pos] = q;
    });
    if (bad) a.read_flag[r] = NPR_ERR_INVALID;
}

constexpr int MEA_RETRY = 1;  // read_flag: the register window was too short for this read, the LDS-ring kernel takes it

// One 64-pair chunk of a read's x-grouped pairs, complete groups only: the lanes find their rank inside their group
// (groups are a few pairs), weigh their pair and change places through LDS so that a group runs from its highest read
// position down (the order the chain kernel wants, see there); sy / sq are rewritten in (x, y) order for the trace.
struct Chunk {
    int valid;      // pairs taken (0: a group of more than 64 pairs)
    uint64_t ends;  // last lane of every group, over the valid lanes
    uint64_t keep;  // lanes whose weight exceeds matchGamma
    int y, q;       // by reference position, and inside a reference position from the highest read position down
    int64_t w;
    int who;        // the pair's place in (x, y) order, relative to the chunk: its id (ties go to the larger)
};
__device__ __forceinline__ Chunk prep_chunk(const MeaArgs &a, int *sx, int *sy, int *sq, int64_t ry, int base, int n, int lane,
                                            int64_t floor_w, int64_t *tw, int *ty, int *tq) {
    Chunk c;
    const int pos = base + lane;
    const bool in = pos < n;
    const int x = in ? sx[pos] : -1, xn = pos + 1 < n ? sx[pos + 1] : -2;
    int y = in ? sy[pos] : 0, q = in ? sq[pos] : 0;
    const uint64_t ends = __ballot(in && x != xn);
    c.valid = ends ? 64 - __builtin_clzll(ends) : 0;
    c.ends = ends, c.keep = 0, c.y = 0, c.q = 0, c.w = 0;
    if (!ends) return c;
    const bool act = lane < c.valid;
    const uint64_t starts = (ends << 1) | 1ull;
    const int gfirst = 63 - __builtin_clzll(starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1)));
    const int gend = act ? lane + __builtin_ctzll(ends >> lane) : lane;
    const int gsize = gend - gfirst + 1;
    int rank = 0, rowsum = q;  // rank by read position inside the group, posterior mass of the reference position
    for (int o = 1; __any(act && o < gsize); ++o) {
        const int lo = lane - o, hi = lane + o;
        const int yl = __shfl(y, lo & 63), ql = __shfl(q, lo & 63), yh = __shfl(y, hi & 63), qh = __shfl(q, hi & 63);
        if (lo >= gfirst) rank += yl < y ? 1 : 0, rowsum += ql;
        if (hi <= gend) rank += yh < y ? 1 : 0, rowsum += qh;
    }
    const int colsum = act ? a.colsum[ry + y] : 0;
    const int64_t gap = max(P1 - rowsum, int64_t(0)) + max(P1 - colsum, int64_t(0));
    const int64_t w0 = q - static_cast<int64_t>(floor(a.gap_gamma * static_cast<double>(gap)));
    wave_sync();
    if (act) {
        const int to = gend - rank;
        tw[to] = w0, ty[to] = y, tq[to] = q;
    }
    wave_sync();
    c.w = tw[lane], c.y = ty[lane], c.q = tq[lane];
    c.keep = __ballot(act && c.w > floor_w);
    c.who = gfirst + gend - lane;
    if (act) sy[base + c.who] = c.y, sq[base + c.who] = c.q;  // the trace reads them back
    return c;
}

// The heaviest chain ending at or below each read position, for the 128 positions [ybase, ybase + 127], in registers:
// position k lives in lane k & 63, register (k >> 6) & 1.  Positions above the window all hold `top`.  A chain is one
// 64-bit key, score << 23 | (last pair + 1): "heavier, ties to the pair that sorts last" is an integer compare (scores
// are not negative: a chain of negative weight beats nothing, the empty chain included, so it is never inserted).
// An insert at position vy raises every held position >= vy the new chain beats -- one compare-and-select per
// register, no loop: the prefix maximum is monotone.  The window moves up when an insert lands above it (the
// positions it leaves are forgotten); a query or insert below the window hands the read to the LDS-ring kernel
// (MEA_RETRY), and so do reads too long for the key (scores from 2^40, pairs from 2^23).
constexpr int WHO_BITS = 23;
__global__ void __launch_bounds__(WAVE) k_mea_chain_win(MeaArgs a) {
    __shared__ int64_t tw[WAVE];
    __shared__ int ty[WAVE], tq[WAVE];
    const int lane = threadIdx.x;
    const int r = a.order[blockIdx.x];
    const int64_t rp = a.rp_off[r], ry = a.ry_off[r];
    const int n = a.read_flag[r] ? 0 : static_cast<int>(a.rp_off[r + 1] - rp);  // (flagged: the scatter was incomplete)
    const int lmin = static_cast<int>(min(a.rx_off[r + 1] - a.rx_off[r] - 1, a.ry_off[r + 1] - ry));
    if (a.ring_only || n >= (1 << WHO_BITS) - 1 || lmin > 100000) {
        if (lane == 0 && a.read_flag[r] == 0) a.read_flag[r] = MEA_RETRY;
        return;
    }
    int *const sx = a.sx + rp, *const sy = a.sy + rp, *const sq = a.sq + rp, *const back = a.back + rp;
    const int64_t floor_w = static_cast<int64_t>(floor(a.match_gamma * static_cast<double>(P1)));
    constexpr int64_t WHO_MASK = (int64_t(1) << WHO_BITS) - 1;
    int ybase = 0;
    int64_t p0 = 0, p1 = 0, top = 0;
    int k0 = lane, k1 = lane + 64;  // positions held
    int flag = 0;
    // heaviest chain over the pairs inserted so far with read position <= key
    auto query = [&](int key) -> int64_t {
        if (static_cast<unsigned>(key - ybase) < 128u) return (key & 64) ? rdlane64(p1, key & 63) : rdlane64(p0, key & 63);  // the usual case
        if (key < 0) return 0;
        if (key > ybase + 127) return top;
        flag = MEA_RETRY;  // below the window
        return 0;
    };
    auto insert = [&](int vy, int64_t total, int who) {
        if (total < 0) return;
        const int64_t v = (total << WHO_BITS) | static_cast<int64_t>(who + 1);
        if (static_cast<unsigned>(vy - ybase) >= 128u) {
            if (vy < ybase) {
                flag = MEA_RETRY;
            } else {  // move the window up: the positions it gains hold the overall maximum
                ybase = vy - 127;
                if (k0 < ybase) p0 = top;
                if (k1 < ybase) p1 = top;
                k0 = ybase + ((lane - ybase) & 127), k1 = ybase + ((lane + 64 - ybase) & 127);
            }
        }
        top = max(top, v);
        if (k0 >= vy) p0 = max(p0, v);
        if (k1 >= vy) p1 = max(p1, v);
    };
    for (int base = 0; base < n && !flag;) {
        const Chunk c = prep_chunk(a, sx, sy, sq, ry, base, n, lane, floor_w, tw, ty, tq);
        if (!c.valid) {
            flag = MEA_RETRY;
            break;
        }
        int bk = -1;
        // The pairs of one reference position must not see each other's chains.  Taken from the highest read position
        // down (the lane order prep_chunk leaves), each can query and insert in one go: an insert at y leaves every
        // position below y alone.
        for (uint64_t todo = c.keep; todo; todo &= todo - 1) {
            const int i = __builtin_ctzll(todo), y = rdlane(c.y, i);
            const int64_t k = query(y - 1);
            if (lane == i) bk = static_cast<int>(k & WHO_MASK) - 1;
            insert(y, rdlane64(c.w, i) + (k >> WHO_BITS), base + rdlane(c.who, i));
        }
        if ((c.keep >> lane) & 1) back[base + c.who] = bk;
        base += c.valid;
    }
    if (lane == 0) {
        a.best_who[r] = static_cast<int>(top & WHO_MASK) - 1;
        if (flag) a.read_flag[r] = flag;
    }
}

// The general version, for the reads k_mea_chain_win gave up on: the prefix maximum in an LDS ring of `ring` read
// positions (score int64, who int32); an insert overwrites the run of entries the new chain beats, all lanes at once.
__global__ void __launch_bounds__(WAVE) k_mea_chain(MeaArgs a) {
    extern __shared__ __align__(16) char lds[];
    const int RING = a.ring, MASK = RING - 1;
    int64_t *const rs = reinterpret_cast<int64_t *>(lds);
    int *const rw = reinterpret_cast<int *>(rs + RING);
    int64_t *const tw = reinterpret_cast<int64_t *>(rw + RING);  // weights in sorted lane order
    int *const ty = reinterpret_cast<int *>(tw + WAVE);
    int *const tq = ty + WAVE;
    const int lane = threadIdx.x;
    const int r = blockIdx.x;
    if (a.read_flag[r] != MEA_RETRY) return;
    const int64_t rp = a.rp_off[r], ry = a.ry_off[r];
    const int n = static_cast<int>(a.rp_off[r + 1] - rp);
    int *const sx = a.sx + rp, *const sy = a.sy + rp, *const sq = a.sq + rp, *const back = a.back + rp;
    const int64_t floor_w = static_cast<int64_t>(floor(a.match_gamma * static_cast<double>(P1)));
    int ytop = -1;  // largest key the ring holds; beyond it the prefix maximum is `top`
    int64_t top_s = 0;
    int top_w = -1;
    int flag = 0;
    for (int base = 0; base < n && !flag;) {
        const Chunk c = prep_chunk(a, sx, sy, sq, ry, base, n, lane, floor_w, tw, ty, tq);
        if (!c.valid) {
            flag = NPR_ERR_CAPACITY;  // more than 64 pairs on one reference position: cannot happen above a 0.01 threshold
            break;
        }
        const int pos = base + c.who, y = c.y;
        const int64_t w = c.w;
        const bool keep = (c.keep >> lane) & 1;
        uint64_t gm = c.ends & (c.valid == 64 ? ~0ull : ((1ull << c.valid) - 1));
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
            wave_sync();
            uint64_t km = __ballot(mine);
            while (km) {
                const int i = __builtin_ctzll(km);
                km &= km - 1;
                const int vy = rdlane(y, i), vw = base + rdlane(c.who, i);
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
                wave_sync();
            }
            gs = ge + 1;
        }
        flag = __any(flag) ? NPR_ERR_CAPACITY : 0;
        base += c.valid;
    }
    if (lane == 0) {
        a.best_who[r] = top_w;
        a.read_flag[r] = flag;
    }
}

// One wavefront per read: the chain's pairs are visited in descending sorted order and mostly a few entries apart,
// so 64 consecutive entries are loaded at once and the back pointers followed from lane to lane; lane 0 writes.
__global__ void __launch_bounds__(WAVE) k_mea_trace(MeaArgs a) {
    const int r = a.order[blockIdx.x], lane = threadIdx.x;
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
        if (hop >= 0 && p > lo) {
            --p;
            if (lane == 0) *p = make_int2(hop, hlen);
        }
        hop = op, hlen = len;
    };
    int cx = lX, cy = lY, len = 0;
    int64_t mass = 0;
    int i = a.read_flag[r] == 0 ? __builtin_amdgcn_readfirstlane(a.best_who[r]) : -1;
    while (i >= 0) {
        const int cb = max(i - (WAVE - 1), 0), idx = min(cb + lane, i);
        const int vb = back[idx], vx = sx[idx], vy = sy[idx], vq = sq[idx];
        // Runs of diagonal steps (the previous pair of the chain is (x-1, y-1): most steps) are found by all lanes at
        // once -- pointer jumping over the back pointers inside the chunk, the posterior mass summed along -- so that
        // the serial walk below takes one step per run, not per pair.
        const int rel = (vb - cb) & (WAVE - 1);
        const bool diag = vb >= cb && __shfl(vx, rel) == vx - 1 && __shfl(vy, rel) == vy - 1;
        int nxt = diag ? rel : lane, val = diag ? vq : 0;  // val: mass of the pairs from this one down to, not including, nxt
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            val += __shfl(val, nxt);
            nxt = __shfl(nxt, nxt);
        }
        while (i >= cb) {
            const int l = i - cb, h = rdlane(nxt, l);
            const int x = rdlane(vx, l), y = rdlane(vy, l), hx = rdlane(vx, h), hy = rdlane(vy, h);
            emit(NPR_OP_I, cy - y - 1);  // backwards: the pair's M comes last in its (D, I, M) triple
            emit(NPR_OP_D, cx - x - 1);
            emit(NPR_OP_M, x - hx + 1);  // the run's pairs
            cx = hx, cy = hy, mass += rdlane(val, l) + rdlane(vq, h), len += x - hx + 1;
            i = rdlane(vb, h);
        }
    }
    emit(NPR_OP_I, cy);
    emit(NPR_OP_D, cx);
    if (hop >= 0 && p > lo) {
        --p;
        if (lane == 0) *p = make_int2(hop, hlen);
    }
    if (lane == 0) {
        a.n_ops[r] = static_cast<int>(reinterpret_cast<int2 *>(a.ops_tmp) + a.ot_off[r + 1] - p);
        a.chain_len[r] = len;
        a.chain_mass[r] = mass;
    }
}

__global__ void __launch_bounds__(256) k_mea_gather(MeaArgs a) {
    for (int r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
        const int n = static_cast<int>(a.od_off[r + 1] - a.od_off[r]);
        const int2 *src = reinterpret_cast<const int2 *>(a.ops_tmp) + a.ot_off[r + 1] - a.n_ops[r];
        uint32_t *dst = a.ops_dense + a.od_off[r];
        for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = static_cast<uint32_t>(src[i].y) << 2 | static_cast<uint32_t>(src[i].x);
    }
}

}  // namespace

size_t mea_chain_lds_bytes(int ring) { return static_cast<size_t>(ring) * 12 + WAVE * (8 + 4 + 4); }

int launch_mea_sort(const MeaArgs &a, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.sort_lds_bytes > 0) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_mea_sort_lds), hipFuncAttributeMaxDynamicSharedMemorySize, a.sort_lds_bytes);
        if (e != hipSuccess) return static_cast<int>(e);
        hipLaunchKernelGGL(k_mea_sort_lds, dim3(a.n_reads), dim3(SORT_THREADS), a.sort_lds_bytes, s, a);
        return static_cast<int>(hipGetLastError());
    }
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
    hipLaunchKernelGGL(k_mea_chain_win, dim3(a.n_reads), dim3(WAVE), 0, s, a);
    hipLaunchKernelGGL(k_mea_chain, dim3(a.n_reads), dim3(WAVE), lds, s, a);  // returns at once unless the read was handed over
    hipLaunchKernelGGL(k_mea_trace, dim3(a.n_reads), dim3(WAVE), 0, s, a);
    return static_cast<int>(hipGetLastError());
}

int launch_mea_gather(const MeaArgs &a, void *stream) {
    const int g = a.n_reads < 8192 ? a.n_reads : 8192;
    hipLaunchKernelGGL(k_mea_gather, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}

}  // namespace npr
