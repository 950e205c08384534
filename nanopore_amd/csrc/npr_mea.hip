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
//   k_mea_weigh    one workgroup per read, a thread per sorted pair: the pair's weight (posterior - gapGamma * gap mass of
//                  its row and column), the pairs not above matchGamma dropped, the kept ones written out twice -- in
//                  (x, y) order (their ids: what back pointers and the trace use) and in the order the chain visits them
//                  (reference positions in order, the pairs of one position from the highest read position down)
//   k_mea_chain_lanes  ONE LANE per read, 64 reads per wavefront (round 4; before: one wavefront per read, every
//                  instruction doing one lane's worth of work -- 21 ms for 4.5e8 pairs, half of the finish): each lane walks
//                  its read's kept pairs; the heaviest chain ending at or below every read position is a monotone prefix
//                  maximum held for a window of 128 read positions in the lane's column of an LDS table; ties go to the
//                  pair that sorts last
//   k_mea_chain    one wavefront per read with the prefix maximum in an LDS ring of any length, over the sorted pairs
//                  themselves (prep_chunk weighs them): the reads whose pairs reach back further than the window (none
//                  on usual data; forced in the tests)
//   k_mea_trace    one wavefront per read: walk the back pointers from the best chain's last pair, writing the ops
//                  backwards (run-length merged) into the read's scratch, and sum the chain's posterior mass
//   k_mea_gather   dense copy of every read's ops, one word each (and the words' low halves when every run of the batch fits 14 bits: what
//                  crosses PCIe then)
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

// the tables of the reads whose spans do not fit the LDS (the others never touch global tables): counts and column sums start from zero
// What this stage reads of a task's pair list: nothing unless its DP pass ended NPR_OK -- a list that overflowed its capacity, or belongs to a task
// that failed, is no input (round 5 lost 25 GPU-minutes to lists that held candidates instead of posteriors; npr_batch_finish gives such a read
// empty tables as well, so either guard alone keeps the stage off them).  And a value that is no probability -- NaN, negative, above 1 -- makes its
// read NPR_ERR_INVALID instead of a weight.
__device__ __forceinline__ int mea_task_pairs(const TaskOut &o, const Task &tk) { return o.status == NPR_OK ? min(max(o.npairs, 0), tk.pair_cap) : 0; }
__device__ __forceinline__ bool mea_is_posterior(float p) { return p >= 0.f && p <= 1.0f + 0x1p-10f; }  // (fp32 rounding puts a certain match at 1 + a few 2^-23: 1.0000013 in configs[1])

__global__ void __launch_bounds__(256) k_mea_zero(MeaArgs a) {
    const int r = blockIdx.x;
    const int64_t c0 = a.cnt_off[r];
    if (c0 < 0) return;
    const int nx = static_cast<int>(a.rx_off[r + 1] - a.rx_off[r]), ny = static_cast<int>(a.ry_off[r + 1] - a.ry_off[r]);
    for (int i = threadIdx.x; i < nx; i += blockDim.x) a.cnt[c0 + i] = 0;
    for (int i = threadIdx.x; i < ny; i += blockDim.x) a.colsum[a.ry_off[r] + i] = 0;
}

__global__ void __launch_bounds__(256) k_mea_count(MeaArgs a) {
    for (int t = blockIdx.x; t < a.ntasks; t += gridDim.x) {
        const Task &tk = a.tasks[t];
        const int n = mea_task_pairs(a.outs[t], tk);
        const int r = tk.read;
        const int64_t rx = a.cnt_off[r], ry = a.ry_off[r];
        if (rx < 0) continue;  // (a read whose tables fit the LDS: k_mea_sort_lds)
        const int lX = static_cast<int>(a.rx_off[r + 1] - a.rx_off[r]) - 1, lY = static_cast<int>(a.ry_off[r + 1] - ry);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int x = a.px[tk.pair_off + i], y = a.py[tk.pair_off + i];
            if (x < 0 || x >= lX || y < 0 || y >= lY || !mea_is_posterior(a.pp[tk.pair_off + i])) {
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
    const int64_t rx = a.cnt_off[r];
    if (rx < 0) return;
    const int n = static_cast<int>(a.rx_off[r + 1] - a.rx_off[r]);  // lX + 1 entries (the last count is zero)
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
        const int n = mea_task_pairs(a.outs[t], tk);
        const int r = tk.read;
        const int64_t rx = a.cnt_off[r], rp = a.rp_off[r];
        if (rx < 0) continue;
        const int lX = static_cast<int>(a.rx_off[r + 1] - a.rx_off[r]) - 1, lY = static_cast<int>(a.ry_off[r + 1] - a.ry_off[r]);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int x = a.px[tk.pair_off + i], y = a.py[tk.pair_off + i];
            if (x < 0 || x >= lX || y < 0 || y >= lY || !mea_is_posterior(a.pp[tk.pair_off + i])) continue;
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
    const int nthreads = static_cast<int>(blockDim.x);  // (SORT_THREADS, or fewer when the launch shares the chip with a DP pass: MeaArgs::sort_threads)
    const int64_t rx = a.rx_off[r], ry = a.ry_off[r], rp = a.rp_off[r];
    const int lX = static_cast<int>(a.rx_off[r + 1] - rx) - 1, lY = static_cast<int>(a.ry_off[r + 1] - ry);
    if (a.rp_off[r + 1] == rp || a.cnt_off[r] >= 0) return;  // (no pairs; or a span beyond the LDS: the three kernels above)
    const int ft = a.read_first[r], nt = a.read_ntasks[r];
    int bad = 0;
    // the pairs of every task of the read, f(x, y, p-quantum)
    auto for_pairs = [&](auto f) {
        for (int s = 0; s < nt; ++s) {
            const int t = a.task_of[ft + s];
            const Task &tk = a.tasks[t];
            const int n = mea_task_pairs(a.outs[t], tk);
            for (int i = tid; i < n; i += nthreads) {
                const int x = a.px[tk.pair_off + i], y = a.py[tk.pair_off + i];
                if (x < 0 || x >= lX || y < 0 || y >= lY || !mea_is_posterior(a.pp[tk.pair_off + i])) {
                    bad = 1;
                    continue;
                }
                f(x, y, static_cast<int>(floor(static_cast<double>(a.pp[tk.pair_off + i]) * static_cast<double>(P1))));
            }
        }
    };
    for (int i = tid; i < lY; i += nthreads) h[i] = 0;
    __syncthreads();
    for_pairs([&](int, int y, int q) { atomicAdd(&h[y], q); });
    __syncthreads();
    for (int i = tid; i < lY; i += nthreads) a.colsum[ry + i] = h[i];
    __syncthreads();
    for (int i = tid; i <= lX; i += nthreads) h[i] = 0;
    __syncthreads();
    for_pairs([&](int x, int, int) { atomicAdd(&h[x], 1); });
    __syncthreads();
    int carry = 0;
    for (int base = 0; base <= lX; base += nthreads) {
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
        for (int k = 0; k < nthreads / WAVE; ++k) before += k < wv ? wsum[k] : 0, total += wsum[k];
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

// Weights of a read's sorted pairs and the two compacted forms of the kept ones.  The pairs of a reference position (a
// "group") are contiguous after the sort, in arbitrary order; a thread looks at its own group only: the group's posterior
// mass (row sum), which of its pairs are kept, where its own pair ranks among them.  ids: the kept pairs in (x, y) order
// (the order ties are decided by: the later pair wins, as in npr_host.cpp's mea_cigar); visit order: groups in order, a
// group from its highest read position down (a pair must not see the chains of its own reference position, and an insert
// at y leaves every position below y alone).
constexpr int WEIGH_CHUNK = 128;  // positions a wavefront weighs per round (two per lane)
constexpr int WEIGH_RING = 512;   // LDS ring of staged positions: the round's chunk, the one before and the one after, and the one being loaded
constexpr int WEIGH_HALO = 64;    // a group (the pairs of one reference position) is looked for this far on both sides
__device__ __forceinline__ int64_t pair_weight(int q, int rowsum, int colsum, double gap_gamma) {
    const int64_t gap = max(P1 - rowsum, int64_t(0)) + max(P1 - colsum, int64_t(0));
    return q - static_cast<int64_t>(floor(gap_gamma * static_cast<double>(gap)));
}
// One wavefront per read, rounds of 128 sorted positions.  The positions are staged in an LDS ring with the column sums of their
// read positions -- the chunk after the round's is fetched while the round is worked on (its x / y / q a round ahead, its column
// sums when they have arrived) --, and everything a lane then does for its two pairs -- the bounds and the posterior mass of
// their groups, which of a group's pairs are kept, the pair's rank among them -- reads LDS.  Many reads are in flight per CU (a
// wavefront and 8 KB of LDS each), which is what hides the latency of a round's one dependent load.  A group that reaches past the
// halo (more than 64 pairs on one reference position: cannot happen above a 0.01 threshold) hands the read to the ring kernel,
// which reports it.
__global__ void __launch_bounds__(WAVE) k_mea_weigh(MeaArgs a) {
    __shared__ int X[WEIGH_RING], Y[WEIGH_RING], Q[WEIGH_RING], C[WEIGH_RING];
    const int r = a.order[blockIdx.x], lane = threadIdx.x;
    const int64_t rp = a.rp_off[r], ry = a.ry_off[r];
    const int n = a.read_flag[r] ? 0 : static_cast<int>(a.rp_off[r + 1] - rp);  // (flagged: the scatter was incomplete)
    const int *const sx = a.sx + rp, *const sy = a.sy + rp, *const sq = a.sq + rp;
    const int *const col = a.colsum + ry;
    const int64_t floor_w = static_cast<int64_t>(floor(a.match_gamma * static_cast<double>(P1)));
    constexpr int RM = WEIGH_RING - 1;
    const int nc = (n + WEIGH_CHUNK - 1) / WEIGH_CHUNK;
    int giveup = 0;
    // registers of the chunk in flight: positions base + lane, base + 64 + lane
    int fx[2], fy[2], fq[2];
    auto fetch = [&](int chunk) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = chunk * WEIGH_CHUNK + h * WAVE + lane;
            const bool ok = j < n;
            fx[h] = ok ? sx[j] : -1, fy[h] = ok ? sy[j] : 0, fq[h] = ok ? sq[j] : 0;  // (-1: no position's group)
        }
    };
    auto stage = [&](int chunk) {  // the fetched chunk and its column sums into the ring
        int fc[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) fc[h] = fx[h] >= 0 ? col[fy[h]] : 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = (chunk * WEIGH_CHUNK + h * WAVE + lane) & RM;
            X[k] = fx[h], Y[k] = fy[h], Q[k] = fq[h], C[k] = fc[h];
        }
    };
    for (int k = lane; k < WEIGH_CHUNK; k += WAVE) X[(-WEIGH_CHUNK + k) & RM] = -2;  // "chunk -1": before the first position
    if (nc > 0) {
        fetch(0);
        stage(0);
        fetch(1);
    }
    int carry = 0;  // kept pairs before this round's positions
    for (int c = 0; c < nc; ++c) {
        stage(c + 1);  // (past the end: sentinels)
        fetch(c + 2);
        wave_sync();
        int keep[2], below[2], above[2], before[2], xs[2], ys[2], qs[2];
        int64_t ws[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = c * WEIGH_CHUNK + h * WAVE + lane, me = i & RM;
            const bool in = i < n;
            const int x = X[me], y = Y[me], q = Q[me];
            int gs = i, ge = i, rowsum = q;
            if (in) {
                while (gs > i - WEIGH_HALO && X[(gs - 1) & RM] == x) rowsum += Q[(--gs) & RM];
                while (ge < i + WEIGH_HALO && X[(ge + 1) & RM] == x) rowsum += Q[(++ge) & RM];
                if (gs == i - WEIGH_HALO || ge == i + WEIGH_HALO) giveup = 1;  // (the group may go on beyond the halo)
            }
            const int64_t w = in ? pair_weight(q, rowsum, C[me], a.gap_gamma) : 0;
            keep[h] = in && w > floor_w;
            // kept pairs of the group: at positions before mine, with a smaller / a larger read position
            before[h] = below[h] = above[h] = 0;
            if (keep[h])
                for (int j = gs; j <= ge; ++j) {
                    const int k = j & RM;
                    if (j != i && pair_weight(Q[k], rowsum, C[k], a.gap_gamma) > floor_w) before[h] += j < i, below[h] += Y[k] < y, above[h] += Y[k] > y;
                }
            xs[h] = x, ys[h] = y, qs[h] = q, ws[h] = w;
        }
        // exclusive prefix count of the kept pairs by position: the first halves of all lanes, then the second halves
        int sc0 = keep[0], sc1 = keep[1];
        for (int o = 1; o < WAVE; o <<= 1) {
            const int t0 = __shfl_up(sc0, o), t1 = __shfl_up(sc1, o);
            if (lane >= o) sc0 += t0, sc1 += t1;
        }
        const int tot0 = __shfl(sc0, WAVE - 1), tot1 = __shfl(sc1, WAVE - 1);
        const int ex[2] = {sc0 - keep[0], tot0 + sc1 - keep[1]};
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (keep[h]) {
                const int64_t g0 = rp + carry + ex[h] - before[h];  // kept pairs before the group
                const int64_t id = g0 + below[h], at = g0 + above[h];
                a.kx[id] = xs[h], a.ky[id] = ys[h], a.kq[id] = qs[h];
                a.vrec[at] = make_int4(ys[h], static_cast<int>(ws[h]), static_cast<int>(id - rp), 0);
            }
        carry += tot0 + tot1;
        wave_sync();
    }
    giveup = __any(giveup);
    if (lane == 0) {
        a.kept[r] = giveup ? -1 : carry;
        if (giveup && a.read_flag[r] == 0) a.read_flag[r] = MEA_RETRY;
    }
}

// Where a read's chain may be cut.  Between the kept pairs i and i + 1 (ids: (x, y) order) the read's chain problem falls apart
// when every pair up to i lies below AND left of every pair from i + 1 on: prefix maximum of y over [0, i] < suffix minimum of y
// over (i, m), and x[i] < x[i + 1].  Every pair behind such a cut dominates every pair before it, so a chain behind it is the
// heaviest chain before it plus a chain of the pairs behind it alone: the pieces are independent problems, the best chain is the
// concatenation of theirs, ties included (a later pair wins a tie, and every pair of a later piece is later).  On nanopore
// posteriors two pairs in three are such cuts.  The host plans np[r] pieces of about 1200 kept pairs; one wavefront per read
// looks for a cut inside the 64-pair block at each planned boundary (forward sweep: prefix maxima at those blocks; backward
// sweep: suffix minima) and writes the boundaries it found, pb[0] = 0 <= pb[1] <= .. <= pb[np] = m; a planned boundary without
// a cut in its block leaves its piece empty (the neighbour takes the pairs).
constexpr int MEA_MAX_PIECES = 64;
__global__ void __launch_bounds__(WAVE) k_mea_cuts(MeaArgs a) {
    __shared__ int PM[MEA_MAX_PIECES][WAVE];
    const int r = a.order[blockIdx.x], lane = threadIdx.x;
    const int np = a.np[r];
    int *const pb = a.pb + a.pboff[r];
    const int m = a.read_flag[r] == 0 ? max(a.kept[r], 0) : 0;
    const int64_t rp = a.rp_off[r];
    const int *const kx = a.kx + rp, *const ky = a.ky + rp;
    const int nb = (m + WAVE - 1) / WAVE;
    if (np <= 1 || nb < 2) {
        for (int k = lane; k <= np; k += WAVE) pb[k] = k == 0 ? 0 : m;  // one piece with everything, the others empty
        return;
    }
    auto target = [&](int j) { return static_cast<int>(static_cast<int64_t>(j) * nb / np); };  // block of the j-th planned boundary
    constexpr int NEG = -(1 << 30), POS = 1 << 30;
    // forward: inclusive prefix maximum of y at the target blocks
    int rm = NEG, j = 1;
    for (int b = 0; b < nb && j < np; ++b) {
        const int idx = b * WAVE + lane;
        const int v = idx < m ? ky[idx] : NEG;
        if (target(j) == b) {
            int before = rm;
            for (int o = 32; o; o >>= 1) before = max(before, __shfl_xor(before, o));
            int sc = v;
            for (int o = 1; o < WAVE; o <<= 1) {
                const int t = __shfl_up(sc, o);
                if (lane >= o) sc = max(sc, t);
            }
            const int pm = max(sc, before);
            while (j < np && target(j) == b) PM[j][lane] = pm, ++j;
        }
        rm = max(rm, v);
    }
    // backward: exclusive suffix minimum at the target blocks; a cut where the prefix maximum lies below it
    int rn = POS;
    j = np - 1;
    for (int b = nb - 1; b >= 0 && j >= 1; --b) {
        const int idx = b * WAVE + lane;
        const int v = idx < m ? ky[idx] : POS;
        if (target(j) == b) {
            int after = rn;
            for (int o = 32; o; o >>= 1) after = min(after, __shfl_xor(after, o));
            int sc = v;  // inclusive suffix minimum
            for (int o = 1; o < WAVE; o <<= 1) {
                const int t = __shfl_down(sc, o);
                if (lane + o < WAVE) sc = min(sc, t);
            }
            int ex = __shfl_down(sc, 1);  // of the lanes above
            if (lane == WAVE - 1) ex = POS;
            const int sm = min(ex, after);
            const bool can = idx + 1 < m && kx[idx] != kx[idx + 1];
            while (j >= 1 && target(j) == b) {
                const uint64_t ok = __ballot(can && PM[j][lane] < sm);
                int at = -1;
                if (ok >> 32) at = 32 + __builtin_ctzll(ok >> 32);
                else if (ok) at = 63 - __builtin_clzll(ok);
                if (lane == 0) pb[j] = at < 0 ? -1 : b * WAVE + at + 1;
                --j;
            }
        }
        rn = min(rn, v);
    }
    wave_sync();
    if (lane == 0) {
        pb[0] = 0, pb[np] = m;
        for (int k = 1; k < np; ++k)
            if (pb[k] < pb[k - 1]) pb[k] = pb[k - 1];  // (no cut in the block, or two planned boundaries in one block)
    }
}

// One LANE per piece of a read (k_mea_cuts), 64 pieces per wavefront.  A chain is one 64-bit key, score << 23 | (id of its last
// pair + 1): "heavier, ties to the pair that sorts last" is an integer compare (scores are not negative: a chain of negative
// weight beats nothing, the empty chain included, so it is never inserted).  M[k & 63] (the lane's column of the LDS table) =
// heaviest chain over the pairs inserted so far with read position <= k, for the 64 positions up to ymax; every position
// above ymax holds `top`.  The prefix maximum is monotone, so an insert at y raises the run of positions from y on that the
// new chain beats and stops at the first it does not.  A query or an insert more than the window below ymax hands the READ to
// the LDS-ring kernel (MEA_RETRY; kept[r] = -1 tells the trace that the read's back pointers are positions among the sorted
// pairs, not ids), and so do reads too long for the key (scores from 2^40, pairs from 2^23).  Nothing hides a step's latency
// in a launch of few wavefronts, so the records are fetched a block of LBLK steps ahead: no memory in the dependent chain but the
// lane's own LDS column.
constexpr int WHO_BITS = 23;
constexpr int LWIN = 64;
constexpr int LBLK = 8;
__global__ void __launch_bounds__(WAVE) k_mea_chain_lanes(MeaArgs a) {
    extern __shared__ __align__(16) char lds_raw[];
    int64_t *const M = reinterpret_cast<int64_t *>(lds_raw) + threadIdx.x;  // M[k * WAVE]: the lane's own banks
    const int lane = threadIdx.x;
    const int slot = blockIdx.x * WAVE + lane;
    const int r = slot < a.n_pieces ? a.lane_read[slot] : -1;
    int m = 0, flag = 0;
    int64_t first = 0;
    int *best = nullptr;
    if (r >= 0) {
        const int j = a.lane_piece[slot];
        best = a.pbest + a.poff[r] + j;
        if (a.read_flag[r] == 0 && a.kept[r] >= 0) {
            const int lmin = static_cast<int>(min(a.rx_off[r + 1] - a.rx_off[r] - 1, a.ry_off[r + 1] - a.ry_off[r]));
            if (a.ring_only || a.rp_off[r + 1] - a.rp_off[r] >= (1 << WHO_BITS) - 1 || lmin > 100000) {
                flag = MEA_RETRY;
            } else {
                const int *const pb = a.pb + a.pboff[r] + j;
                first = a.rp_off[r] + pb[0];
                m = pb[1] - pb[0];
            }
        }
    }
    const int4 *const vrec = a.vrec + first;
    int *const kback = a.kback + (r >= 0 ? a.rp_off[r] : 0);
    constexpr int64_t WHO_MASK = (int64_t(1) << WHO_BITS) - 1;
    int mmax = m;
    for (int o = 32; o; o >>= 1) mmax = max(mmax, __shfl_xor(mmax, o));
    int ymax = -1;
    int64_t top = 0;
    int4 nxt[LBLK];
#pragma unroll
    for (int k = 0; k < LBLK; ++k) nxt[k] = k < m ? vrec[k] : make_int4(0, 0, 0, 0);
    for (int t0 = 0; t0 < mmax; t0 += LBLK) {
        int4 cur[LBLK];
#pragma unroll
        for (int k = 0; k < LBLK; ++k) cur[k] = nxt[k];
#pragma unroll
        for (int k = 0; k < LBLK; ++k) nxt[k] = t0 + LBLK + k < m ? vrec[t0 + LBLK + k] : make_int4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < LBLK; ++k) {
            const int y = cur[k].x, w = cur[k].y, id = cur[k].z;
            if (t0 + k < m && !flag) {
                // heaviest chain over the pairs inserted so far with read position < y
                const int key = y - 1;
                int64_t kk = 0;
                if (key > ymax) kk = top;
                else if (key >= 0) {
                    if (key <= ymax - LWIN) flag = MEA_RETRY;
                    else kk = M[(key & (LWIN - 1)) * WAVE];
                }
                kback[id] = static_cast<int>(kk & WHO_MASK) - 1;
                const int64_t total = w + (kk >> WHO_BITS);
                if (total >= 0 && !flag) {
                    const int64_t v = (total << WHO_BITS) | static_cast<int64_t>(id + 1);
                    if (y > ymax) {  // the positions in between hold what every position above ymax holds
                        for (int p = max(ymax + 1, y - (LWIN - 1)); p < y; ++p) M[(p & (LWIN - 1)) * WAVE] = top;
                        top = max(top, v);
                        M[(y & (LWIN - 1)) * WAVE] = top;
                        ymax = y;
                    } else if (y <= ymax - LWIN) {
                        flag = MEA_RETRY;
                    } else {
                        for (int p = y; p <= ymax; ++p) {
                            int64_t *const e = M + (p & (LWIN - 1)) * WAVE;
                            if (*e >= v) break;
                            *e = v;
                        }
                        top = max(top, v);
                    }
                }
            }
        }
    }
    if (r >= 0) {
        *best = static_cast<int>(top & WHO_MASK) - 1;
        if (flag) a.read_flag[r] = flag, a.kept[r] = -1;  // (every piece of the read may write this: the same values)
    }
}

// The general version, for the reads k_mea_chain_lanes gave up on: the prefix maximum in an LDS ring of `ring` read
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
    // (a fixed grid over the reads: a block per read, each with the ring's 100 KB of LDS, cost 1 ms of launches that return at once)
    for (int r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
    if (a.read_flag[r] != MEA_RETRY) continue;
    wave_sync();
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
}

// One wavefront per read: the chain's pairs are visited in descending sorted order and mostly a few entries apart,
// so 64 consecutive entries are loaded at once and the back pointers followed from lane to lane; lane 0 writes.
__global__ void __launch_bounds__(WAVE) k_mea_trace(MeaArgs a) {
    const int r = a.order[blockIdx.x], lane = threadIdx.x;
    const int64_t rp = a.rp_off[r];
    // the kept pairs by id (k_mea_chain_lanes), or the sorted pairs themselves for a read the ring kernel took over
    const bool by_id = __builtin_amdgcn_readfirstlane(a.kept[r]) >= 0;
    const int *sx = (by_id ? a.kx : a.sx) + rp, *sy = (by_id ? a.ky : a.sy) + rp, *sq = (by_id ? a.kq : a.sq) + rp, *back = (by_id ? a.kback : a.back) + rp;
    const int lX = static_cast<int>(a.rx_off[r + 1] - a.rx_off[r]) - 1, lY = static_cast<int>(a.ry_off[r + 1] - a.ry_off[r]);
    int2 *const lo = reinterpret_cast<int2 *>(a.ops_tmp) + a.ot_off[r];
    int2 *p = reinterpret_cast<int2 *>(a.ops_tmp) + a.ot_off[r + 1];  // filled from the end
    int hop = -1, hlen = 0, longest = 0;
    auto emit = [&](int op, int len) {
        if (len <= 0) return;
        if (op == hop) {
            hlen += len;
            return;
        }
        if (hop >= 0 && p > lo) {
            --p;
            if (lane == 0) *p = make_int2(hop, hlen);
            longest = max(longest, hlen);
        }
        hop = op, hlen = len;
    };
    int cx = lX, cy = lY, len = 0;
    int64_t mass = 0;
    bool broken = false;
    // the pieces of the read from the last to the first, each from its heaviest chain's last pair down (by_id); one chain else
    const bool alive = a.read_flag[r] == 0;
    int piece = by_id ? __builtin_amdgcn_readfirstlane(a.np[r]) : 1;
    const int *const pbest = a.pbest + (by_id ? a.poff[r] : 0);
    while (piece-- > 0) {
    int i = !alive ? -1 : __builtin_amdgcn_readfirstlane(by_id ? pbest[piece] : a.best_who[r]);
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
            const int to = rdlane(vb, h);
            if (to >= i) {  // a chain's back pointers go DOWN the list: anything else is a malformed table, and following it need not end
                broken = true;
                i = -1;
                break;
            }
            i = to;
        }
    }
    if (broken) break;
    }
    emit(NPR_OP_I, cy);
    emit(NPR_OP_D, cx);
    if (hop >= 0 && p > lo) {
        --p;
        if (lane == 0) *p = make_int2(hop, hlen);
        longest = max(longest, hlen);
    }
    if (lane == 0) {
        if (broken) a.read_flag[r] = NPR_ERR_INVALID;
        a.max_run[r] = longest;
        a.n_ops[r] = broken ? 0 : static_cast<int>(reinterpret_cast<int2 *>(a.ops_tmp) + a.ot_off[r + 1] - p);
        a.chain_len[r] = len;
        a.chain_mass[r] = mass;
    }
}

__global__ void __launch_bounds__(256) k_mea_gather(MeaArgs a) {
    for (int r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
        const int n = static_cast<int>(a.od_off[r + 1] - a.od_off[r]);
        const int2 *src = reinterpret_cast<const int2 *>(a.ops_tmp) + a.ot_off[r + 1] - a.n_ops[r];
        uint32_t *dst = a.ops_dense + a.od_off[r];
        uint16_t *const dst16 = a.ops_dense16 ? a.ops_dense16 + a.od_off[r] : nullptr;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const uint32_t w = static_cast<uint32_t>(src[i].y) << 2 | static_cast<uint32_t>(src[i].x);
            dst[i] = w;
            if (dst16) dst16[i] = static_cast<uint16_t>(w);
        }
    }
}

}  // namespace

size_t mea_chain_lds_bytes(int ring) { return static_cast<size_t>(ring) * 12 + WAVE * (8 + 4 + 4); }

int launch_mea_sort(const MeaArgs &a, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.sort_lds_bytes > 0) {  // the reads whose per-position tables fit the LDS
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_mea_sort_lds), hipFuncAttributeMaxDynamicSharedMemorySize, a.sort_lds_bytes);
        if (e != hipSuccess) return static_cast<int>(e);
        hipLaunchKernelGGL(k_mea_sort_lds, dim3(a.n_reads), dim3(a.sort_threads >= WAVE && a.sort_threads < SORT_THREADS ? a.sort_threads & ~(WAVE - 1) : SORT_THREADS), a.sort_lds_bytes, s, a);
    }
    if (a.any_global_sort) {  // the others (a read of more than 16 k bases among 12 000 shorter ones used to send ALL of them this way)
        const int tg = a.ntasks < 8192 ? (a.ntasks > 0 ? a.ntasks : 1) : 8192;
        hipLaunchKernelGGL(k_mea_zero, dim3(a.n_reads), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_mea_count, dim3(tg), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_mea_scan, dim3(a.n_reads), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_mea_scatter, dim3(tg), dim3(256), 0, s, a);
    }
    return static_cast<int>(hipGetLastError());
}

int launch_mea_chain(const MeaArgs &a, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = mea_chain_lds_bytes(a.ring);
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_mea_chain), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return static_cast<int>(e);
    const int lanes_lds = LWIN * WAVE * static_cast<int>(sizeof(int64_t));
    const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(k_mea_chain_lanes), hipFuncAttributeMaxDynamicSharedMemorySize, lanes_lds);
    if (e2 != hipSuccess) return static_cast<int>(e2);
    hipLaunchKernelGGL(k_mea_weigh, dim3(a.n_reads), dim3(WAVE), 0, s, a);
    hipLaunchKernelGGL(k_mea_cuts, dim3(a.n_reads), dim3(WAVE), 0, s, a);
    hipLaunchKernelGGL(k_mea_chain_lanes, dim3((a.n_pieces + WAVE - 1) / WAVE), dim3(WAVE), lanes_lds, s, a);
    hipLaunchKernelGGL(k_mea_chain, dim3(a.n_reads < 512 ? (a.n_reads > 0 ? a.n_reads : 1) : 512), dim3(WAVE), lds, s, a);  // (looks at the reads that were handed over)
    hipLaunchKernelGGL(k_mea_trace, dim3(a.n_reads), dim3(WAVE), 0, s, a);
    return static_cast<int>(hipGetLastError());
}

int launch_mea_gather(const MeaArgs &a, void *stream) {
    const int g = a.n_reads < 8192 ? a.n_reads : 8192;
    hipLaunchKernelGGL(k_mea_gather, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return static_cast<int>(hipGetLastError());
}

}  // namespace npr
