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
constexpr int BAND_SUB = 4;                          // anti-diagonals per thread and tile
constexpr int BAND_TILE = BAND_THREADS * BAND_SUB;   // anti-diagonals per tile
constexpr int BAND_STAGE = BAND_TILE;                // plan points of a tile staged in LDS (a piece owns at least one anti-diagonal, as a rule)

// One workgroup per segment, tiles of 1024 anti-diagonals.  The plan points a tile can fall into are staged in LDS first (one coalesced
// load), so that every row's search for its piece reads LDS: rounds 2-4 searched global memory per row, in tiles of 256 with four barriers
// each -- 3 ms for 6 250 reads on an idle chip and 38 ms in the wavefront slots a running DP pass leaves (every dependent load at the
// latency of a loaded memory system), which is what a pipelined job's next DP pass waited for.  Same rows, same
// summary as before: npr_batch_plan_check compares them with the host planner's.
__global__ void __launch_bounds__(BAND_THREADS) k_plan_bands(PlanArgs a) {
    __shared__ long long s_cells[BAND_THREADS], s_gen[BAND_THREADS];
    __shared__ int s_w[BAND_THREADS], s_bad[BAND_THREADS], s_rough[BAND_THREADS];
    __shared__ int s_k;                                   // piece of the tile's first anti-diagonal
    __shared__ int s_lo[BAND_TILE], s_hi[BAND_TILE];      // this tile's rows ...
    __shared__ int s_carry[2][2];                         // ... and the last row of the tile before (by the tile's parity)
    __shared__ PlanPoint s_pt[BAND_STAGE + 1];
    const int tid = threadIdx.x;
    for (int g = blockIdx.x; g < a.n_segs; g += gridDim.x) {
        const PlanSeg sg = a.segs[g];
        const PlanPoint *P = a.points + sg.point_first;
        const int D = sg.lX + sg.lY, m = sg.pieces;
        long long cells = 0, gen = 0;
        int wmax = 0, bad = 0, rough = 0;
        if (tid == 0) s_k = band_piece(P, m, 0);
        __syncthreads();
        int par = 0;
        for (int d0 = 0; d0 <= D; d0 += BAND_TILE, par ^= 1) {
            const int k0 = s_k;
            const int staged = min(BAND_STAGE + 1, m + 1 - k0);  // points k0 .. k0 + staged - 1 (P has m + 1 points)
            for (int i = tid; i < staged; i += BAND_THREADS) s_pt[i] = P[k0 + i];
            __syncthreads();
            auto pt = [&](int k) -> PlanPoint { return k - k0 < staged ? s_pt[k - k0] : P[k]; };
            const int kmax = min(k0 + staged - 2, m - 1);  // last piece whose end point is staged too
            int klast = k0;
#pragma unroll
            for (int j = 0; j < BAND_SUB; ++j) {
                const int q = j * BAND_THREADS + tid, d = d0 + q;
                if (d <= D) {
                    int lo = k0, hi = kmax;  // the last k in [k0, kmax] with d0(P_k) <= d
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (s_pt[mid - k0].d0() <= d) lo = mid; else hi = mid - 1;
                    }
                    if (lo == kmax && kmax < m - 1 && pt(kmax + 1).d0() <= d) lo = band_piece(P, m, d);  // (pieces without a row: beyond what is staged)
                    const BandRow r = band_row_of_piece(a.fixed_mode, a.width, sg.lX, sg.lY, pt(lo), pt(lo + 1), d);
                    a.lo[sg.band_off + d] = r.lo;
                    a.n[sg.band_off + d] = r.n;
                    cells += r.n > 0 ? r.n : 0;
                    gen += r.n > 0 ? ((r.n + 3) & ~3) : 0;
                    wmax = max(wmax, r.n);
                    bad += r.n < 1;
                    s_lo[q] = r.lo, s_hi[q] = r.lo + 2 * (r.n - 1);
                    klast = lo;
                }
            }
            if (tid == BAND_THREADS - 1 && d0 + BAND_TILE <= D) {  // on to the next tile: its first row's piece, this tile's last row
                int k = klast;
                while (k + 1 < m && pt(k + 1).d0() <= d0 + BAND_TILE) ++k;
                s_k = k;
                s_carry[par][0] = s_lo[BAND_TILE - 1], s_carry[par][1] = s_hi[BAND_TILE - 1];
            }
            __syncthreads();
            // both edges must move by exactly one cell per anti-diagonal (what the stripe table's binary searches and the
            // frame schedules rely on); a band that does not is flagged and takes the general paths
#pragma unroll
            for (int j = 0; j < BAND_SUB; ++j) {
                const int q = j * BAND_THREADS + tid, d = d0 + q;
                if (d <= D && d > 0) {
                    const int pl = q ? s_lo[q - 1] : s_carry[par ^ 1][0], ph = q ? s_hi[q - 1] : s_carry[par ^ 1][1];
                    const int dl = s_lo[q] - pl, dh = s_hi[q] - ph;
                    rough += (dl != 1 && dl != -1) || (dh != 1 && dh != -1);
                }
            }
        }
        s_cells[tid] = cells, s_gen[tid] = gen, s_w[tid] = wmax, s_bad[tid] = bad, s_rough[tid] = rough;
        __syncthreads();
        for (int k = BAND_THREADS / 2; k > 0; k >>= 1) {
            if (tid < k) {
                s_cells[tid] += s_cells[tid + k], s_gen[tid] += s_gen[tid + k];
                s_w[tid] = max(s_w[tid], s_w[tid + k]), s_bad[tid] += s_bad[tid + k];
                s_rough[tid] += s_rough[tid + k];
            }
            __syncthreads();
        }
        if (tid == 0) a.summary[g] = SegSummary{s_cells[0], s_gen[0], s_w[0], s_bad[0], s_rough[0], 0};
        __syncthreads();
    }
}

// The frame schedule is a sequential scan per segment -- every rebase depends on the frame's position, which depends on every
// step before it -- and rounds 1-3 walked it that way (k_plan_sched: a lane per segment, eight segments per wavefront, rows staged
// through LDS): 14 ms for the 24 576 segments of the headline batch and 20 ms for ANY batch that holds one 20 kb read, because the
// walk of the longest segment (40 000 steps of ~1 000 cycles) is what the staging waits for.  Round 4 cuts the walk into chunks:
//
//   The only state a step depends on is flo, the frame's first x-y (npr_sched.h stair_step), and on the positions a valid
//   schedule can be in a step is a CLAMP of it: an X-step takes flo to max(flo + 1, c) with c = max(hi - span, hi_next - span + 1)
//   (rebase when the band's top would leave the frame now, or at the Y-step after it, which cannot rebase upwards), a Y-step to
//   min(flo - 1, c) with c = min(lo, lo_next - 1).  Clamps of a shifted argument compose to a clamp of a shifted argument:
//   SCHED_CHUNK steps take flo to min(max(flo + A, L), U).
//
//   k_sched_compose  a lane per chunk: (A, L, U) of its steps -- no state, a few integer operations per step
//   k_sched_starts   a lane per segment: flo at the head of each of its chunks, chunk after chunk (a few dozen clamps)
//   k_sched_walk     a lane per chunk: the real walk (stair_step: the host planner's function) from that flo, control words with the
//                    chunk's own row offsets; it goes on two steps into the next chunk to learn their `moved` bits (which need the
//                    band of the two rows before)
//   k_sched_finish   a lane per segment: prefix of the chunks' cells (the row offsets' base), whether every chunk could be followed and the
//                    schedule fits its class -> the segment's class, or on to its next candidate class
//   k_sched_patch    a wavefront per chunk: the offset base added to its words, the first two rows' `moved` bits set
//
// once per class that is a candidate of any segment, smallest frame first.  A chunk whose walk leaves the valid positions (ok = 0) fails
// the segment for that class exactly as the sequential walk would: the clamp form only differs from stair_step where that one fails.
// npr_batch_plan_check compares the result word for word with the host planner's sequential stair_schedule (tests/test_gpu_plan.py).
constexpr int SCHED_CHUNK = 256;
struct SchedChunk {
    int32_t A, L, U;       // the chunk's steps as a map of flo
    int32_t flo;           // at its head (before its first step)
    uint32_t cells;        // rows of its steps, in scratch cells
    int32_t ok;
    uint32_t moved_next;   // bit 0 / 1: the `moved` bits of the first / second row of the NEXT chunk
    uint32_t base;         // scratch cells before the chunk
    int32_t seg;           // the segment it belongs to
    int32_t pad[3];
};
struct SchedWork {
    SchedArgs a;
    const int64_t *chunk_off;  // per segment: its first chunk (prefix sum of the chunks per segment; n_segs + 1 entries)
    SchedChunk *chunks;
    int32_t *cur;              // per segment: the class being tried, or -1 when it has its class / no candidate is left
    int64_t n_chunks;
    int32_t cls;               // the class of this round
    int32_t *active;           // [kSchedClasses]: segments that try class c (a round without any returns at once)
};
struct RowQuad {
    int lo[4], n[4];
};
__device__ __forceinline__ RowQuad row_quad(const int32_t *lo_, const int32_t *n_, int d) {  // rows d .. d + 3 (dword-aligned 16-byte loads)
    typedef int v4 __attribute__((ext_vector_type(4), aligned(4)));
    const v4 a = *reinterpret_cast<const v4 *>(lo_ + d), b = *reinterpret_cast<const v4 *>(n_ + d);
    return RowQuad{{a.x, a.y, a.z, a.w}, {b.x, b.y, b.z, b.w}};
}
__global__ void __launch_bounds__(64) k_sched_begin(SchedWork w) {  // the first candidate class of every segment
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= w.a.n_segs) return;
    const uint32_t cand = (w.a.ctl_off[g] >= 0 && w.a.summary[g].bad == 0) ? w.a.cand[g] : 0u;
    w.cur[g] = cand ? __builtin_ctz(cand) : -1;
    if (cand) atomicAdd(w.active + __builtin_ctz(cand), 1);
    w.a.cls[g] = -1, w.a.cells[g] = 0;
    for (int64_t ci = w.chunk_off[g]; ci < w.chunk_off[g + 1]; ++ci) w.chunks[ci].seg = g;
}
__global__ void __launch_bounds__(64) k_sched_compose(SchedWork w) {
    if (w.active[w.cls] == 0) return;
    const int64_t ci = static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x;
    if (ci >= w.n_chunks) return;
    const int g = w.chunks[ci].seg;
    if (w.cur[g] != w.cls) return;
    const PlanSeg sg = w.a.segs[g];
    const int D = sg.lX + sg.lY, C = 64 * kSchedR[w.cls] * kSchedNW[w.cls], span = 2 * (C - 1);
    const int d0 = static_cast<int>(ci - w.chunk_off[g]) * SCHED_CHUNK, d1 = min(d0 + SCHED_CHUNK - 1, D);
    const int32_t *lo_ = w.a.lo + sg.band_off, *n_ = w.a.n + sg.band_off;
    constexpr int BIG = 1 << 29;
    int A = 0, L = -BIG, U = BIG;
    // rows four at a time (a lane's rows are its own cache lines: a 16-byte load per array and four steps instead of four 4-byte ones;
    // d_lo / d_n are padded so that the look-ahead past a segment's last row stays inside them), the next four fetched ahead
    RowQuad cur = row_quad(lo_, n_, d0), nxt = row_quad(lo_, n_, d0 + 4);
    for (int q0 = d0; q0 <= d1; q0 += 4) {
        const RowQuad far = row_quad(lo_, n_, q0 + 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = q0 + i;
            if (d > d1) break;
            const bool next = d < D;
            const int lo_c = cur.lo[i], n_c = cur.n[i];
            const int lo_x = i < 3 ? cur.lo[i + 1] : nxt.lo[0], n_x = i < 3 ? cur.n[i + 1] : nxt.n[0];
            if (d > 0) {
                if (d & 1) {
                    const int hi = lo_c + 2 * (n_c - 1), hi_nx = lo_x + 2 * (n_x - 1);
                    const int c = max(hi - span, next ? hi_nx - span + 1 : -BIG);
                    A += 1, L = max(L + 1, c), U = max(U + 1, c);
                } else {
                    const int c = min(lo_c, next ? lo_x - 1 : BIG);
                    A -= 1, L = min(L - 1, c), U = min(U - 1, c);
                }
            }
        }
        cur = nxt, nxt = far;
    }
    SchedChunk &k = w.chunks[ci];
    k.A = A, k.L = L, k.U = U;
}
__global__ void __launch_bounds__(64) k_sched_starts(SchedWork w) {
    if (w.active[w.cls] == 0) return;
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= w.a.n_segs || w.cur[g] != w.cls) return;
    const PlanSeg sg = w.a.segs[g];
    const int R = kSchedR[w.cls], NW = kSchedNW[w.cls];
    StairState st{0, 0, 0, 0, 0, 0};
    const bool ok = stair_begin(st, w.a.lo[sg.band_off], w.a.n[sg.band_off], w.a.summary[g].max_width, R, NW);
    int flo = st.flo;
    for (int64_t ci = w.chunk_off[g]; ci < w.chunk_off[g + 1]; ++ci) {
        SchedChunk &k = w.chunks[ci];
        k.flo = flo, k.ok = ok ? 1 : 0;
        flo = min(max(flo + k.A, k.L), k.U);
    }
}
__global__ void __launch_bounds__(64) k_sched_walk(SchedWork w) {
    if (w.active[w.cls] == 0) return;
    const int64_t ci = static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x;
    if (ci >= w.n_chunks) return;
    const int g = w.chunks[ci].seg;
    if (w.cur[g] != w.cls) return;
    SchedChunk &k = w.chunks[ci];
    if (!k.ok) return;  // (stair_begin refused the class)
    const PlanSeg sg = w.a.segs[g];
    const int R = kSchedR[w.cls], NW = kSchedNW[w.cls], rshift = stair_rshift(R), C = 64 * R * NW;
    const int D = sg.lX + sg.lY;
    const int d0 = static_cast<int>(ci - w.chunk_off[g]) * SCHED_CHUNK, d1 = min(d0 + SCHED_CHUNK - 1, D);
    const int32_t *lo_ = w.a.lo + sg.band_off, *n_ = w.a.n + sg.band_off;
    uint2 *ctl = reinterpret_cast<uint2 *>(w.a.ctl + 2 * w.a.ctl_off[g]);
    StairState st{k.flo, 0, 0, 0, 0, 0};
    int ok = 1;
    typedef unsigned v4u __attribute__((ext_vector_type(4), aligned(8)));
    uint32_t cells = 0, moved_next = 0;
    const int dlast = min(d1 + 2, D);  // two rows into the next chunk: their `moved` bits need this chunk's last two bands
    RowQuad cur = row_quad(lo_, n_, d0), nxt = row_quad(lo_, n_, d0 + 4);
    for (int q0 = d0; q0 <= dlast && ok; q0 += 4) {
        const RowQuad far = row_quad(lo_, n_, q0 + 8);
        uint32_t w0[4], w1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = q0 + i;
            w0[i] = w1[i] = 0;
            if (d > dlast || !ok) continue;
            const int lo_x = i < 3 ? cur.lo[i + 1] : nxt.lo[0], n_x = i < 3 ? cur.n[i + 1] : nxt.n[0];
            ok = stair_step(st, d, D, cur.lo[i], cur.n[i], d < D ? lo_x : 0, d < D ? n_x : 0, rshift, C, w0[i], w1[i]) ? 1 : 0;
            if (d <= d1) {
                cells = st.off;
            } else {
                moved_next |= ((w1[i] >> 30) & 1u) << (d - d1 - 1);
                ok = 1;  // (what goes wrong in the next chunk's rows is that chunk's to report)
            }
        }
        // the four rows' words (a chunk's rows are whole quads but the segment's last; a failed chunk's words are never looked at)
#pragma unroll
        for (int i = 0; i < 4; i += 2)
            if (q0 + i + 1 <= d1) *reinterpret_cast<v4u *>(ctl + q0 + i) = v4u{w0[i], w1[i], w0[i + 1], w1[i + 1]};
            else if (q0 + i <= d1) ctl[q0 + i] = make_uint2(w0[i], w1[i]);
        cur = nxt, nxt = far;
    }
    k.cells = cells, k.ok = ok, k.moved_next = moved_next;
}
__global__ void __launch_bounds__(64) k_sched_finish(SchedWork w) {
    if (w.active[w.cls] == 0) return;
    const int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= w.a.n_segs || w.cur[g] != w.cls) return;
    const bool packed = stair_packed(kSchedR[w.cls], kSchedNW[w.cls]);
    uint64_t total = 0;
    bool ok = true;
    for (int64_t ci = w.chunk_off[g]; ci < w.chunk_off[g + 1]; ++ci) {
        SchedChunk &k = w.chunks[ci];
        k.base = static_cast<uint32_t>(total);
        ok = ok && k.ok != 0;
        total += k.cells;
        if (total >= (packed ? uint64_t((1u << 29) - 512u) : (uint64_t(1) << 32))) ok = false;  // (stair_step's limit on the row offsets)
    }
    if (ok) {
        w.a.cls[g] = w.cls, w.a.cells[g] = static_cast<int64_t>(total);
        w.cur[g] = -1 - w.cls - 1;  // done in this round: k_sched_patch finishes its words (-2 - cls), then nothing looks at it again
    } else {
        const uint32_t left = w.a.cand[g] & ~((2u << w.cls) - 1u);
        w.cur[g] = left ? __builtin_ctz(left) : -1;
        if (left) atomicAdd(w.active + __builtin_ctz(left), 1);
    }
}
__global__ void __launch_bounds__(64) k_sched_patch(SchedWork w) {
    if (w.active[w.cls] == 0) return;
    const int64_t ci = blockIdx.x;
    const int lane = threadIdx.x;
    const int g = w.chunks[ci].seg;
    if (w.cur[g] != -2 - w.cls) return;
    const PlanSeg sg = w.a.segs[g];
    const int D = sg.lX + sg.lY;
    const int64_t c0 = w.chunk_off[g];
    const int d0 = static_cast<int>(ci - c0) * SCHED_CHUNK, d1 = min(d0 + SCHED_CHUNK - 1, D);
    const bool packed = stair_packed(kSchedR[w.cls], kSchedNW[w.cls]);
    const uint32_t add = packed ? w.chunks[ci].base << 3 : w.chunks[ci].base;
    const uint32_t mv = ci > c0 ? w.chunks[ci - 1].moved_next : 0u;
    uint2 *ctl = reinterpret_cast<uint2 *>(w.a.ctl + 2 * w.a.ctl_off[g]);
    for (int d = d0 + lane; d <= d1; d += 64) {
        uint2 v = ctl[d];
        v.x += add;
        if (ci > c0 && d - d0 < 2) v.y = (v.y & ~(1u << 30)) | (((mv >> (d - d0)) & 1u) << 30);
        ctl[d] = v;
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
size_t plan_sched_chunk_bytes(int64_t n_chunks) { return sizeof(SchedChunk) * static_cast<size_t>(n_chunks > 0 ? n_chunks : 1); }
int64_t plan_sched_chunks_of(int64_t D) { return D / SCHED_CHUNK + 1; }  // chunks of a segment with D + 1 anti-diagonals
int launch_plan_sched(const SchedArgs &a, const int64_t *chunk_off, int64_t n_chunks, void *chunks, int32_t *cur, uint32_t cand_union, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    SchedWork w{a, chunk_off, static_cast<SchedChunk *>(chunks), cur, n_chunks, 0, cur + a.n_segs};  // (cur: n_segs + kSchedClasses ints)
    if (hipMemsetAsync(w.active, 0, sizeof(int32_t) * kSchedClasses, s) != hipSuccess) return static_cast<int>(hipGetLastError());
    const int gs = (a.n_segs + 63) / 64 > 0 ? (a.n_segs + 63) / 64 : 1;
    const int gc = static_cast<int>((n_chunks + 63) / 64 > 0 ? (n_chunks + 63) / 64 : 1);
    hipLaunchKernelGGL(k_sched_begin, dim3(gs), dim3(64), 0, s, w);
    for (int c = 0; c < kSchedClasses; ++c) {
        if (!((cand_union >> c) & 1u)) continue;
        w.cls = c;
        hipLaunchKernelGGL(k_sched_compose, dim3(gc), dim3(64), 0, s, w);
        hipLaunchKernelGGL(k_sched_starts, dim3(gs), dim3(64), 0, s, w);
        hipLaunchKernelGGL(k_sched_walk, dim3(gc), dim3(64), 0, s, w);
        hipLaunchKernelGGL(k_sched_finish, dim3(gs), dim3(64), 0, s, w);
        hipLaunchKernelGGL(k_sched_patch, dim3(static_cast<unsigned>(n_chunks > 0 ? n_chunks : 1)), dim3(64), 0, s, w);
    }
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
