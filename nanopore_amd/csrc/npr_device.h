// npr_device.h -- structures shared between the host API and the HIP kernels of libnprealign.
#pragma once
#include <cstdint>

#include "npr_internal.h"
#include "npr_band.h"

namespace npr {

// One banded DP problem on the device (a Segment of a read).
struct Task {
    int64_t x_off;     // first reference base code of the segment in d_seq
    int64_t y_off;     // first read base code of the segment in d_seq
    int64_t band_off;  // first anti-diagonal entry in d_lo / d_n / d_coff
    int64_t pair_off;  // first slot of this task in the posterior pair buffers
    int32_t lX, lY;    // segment spans
    int32_t D;         // lX + lY
    int32_t pair_cap;  // slots available at pair_off
    int32_t flags;     // bit0 ragged start, bit1 ragged end
    int32_t model;     // model slot
    int32_t xs, ys;    // offset of the segment inside the read's slice (added to emitted coordinates)
    int32_t read;      // owning read
    int32_t cells_pad; // padded cell count (scratch use)
    int64_t ctl_off;   // register-kernel tasks: first anti-diagonal entry in d_ctl (pairs of words), else -1
    int64_t tile_off;  // stripe-kernel tasks: index of the task's header in d_stripes (Stripe units), else -1
    int64_t rowmask_off;  // stripe-kernel tasks: first row of the task in d_rowmask (one word per row of a stripe), else -1
};

// k_dp_tile cuts a task's lattice columns into stripes of at most 64*R columns; one wavefront sweeps a stripe
// anti-diagonal by anti-diagonal, one row of the forward scratch per anti-diagonal.  A task's table is a header
// {X = number of stripes, K = rows of the whole task} followed by its stripes in column order.
struct Stripe {
    int32_t X;      // first lattice column of the stripe
    int32_t K;      // lattice columns (a multiple of the slots per lane; the last stripe may reach past lX)
    int32_t df, dl; // first / last anti-diagonal on which the band has cells in the stripe (dl < df: none)
    uint32_t row0;  // row index of anti-diagonal df in the task's scratch; anti-diagonal d is row row0 + d - df
    int32_t pad[3];
};

struct TaskOut {
    float tot_m;      // total probability = tot_m * 2^tot_e (forward)
    int32_t tot_e;
    float btot_m;     // same from the backward pass
    int32_t btot_e;
    int32_t npairs;   // pairs >= threshold found (may exceed pair_cap -> status NPR_ERR_CAPACITY)
    int32_t status;
};

struct KernelArgs {
    const Task *tasks;
    TaskOut *outs;
    int32_t *queue;  // work-queue head
    int32_t ntasks;
    const DevModel *models;
    const uint8_t *seq;
    const int32_t *lo;
    const int32_t *n;
    const uint32_t *coff;  // per anti-diagonal: offset of its first cell inside the task (cells padded to x4)
    const Stripe *stripes;  // k_dp_tile: stripe tables (Task::tile_off)
    const uint32_t *rowmask;  // k_dp_tile: lane masks of every row, packed (npr_sched.h tile_row_word; Task::rowmask_off)
    unsigned long long *prof;  // k_dp_tile, NPR_TILE_PROF=1: wait-cycle counters
    const int64_t *region;  // k_dp_tile: first scratch cell of each workgroup (regions sized by the workgroup's first task)
    const uint32_t *ctl;   // register kernel: two control words per anti-diagonal (row offset; jlo | n << 13 | (rebase + 1) << 26)
    char *F;               // forward match-state scratch: one region of 8*slot_stride bytes per resident wave.  The
                           // register kernel keeps (mantissa, exponent) interleaved per cell; the generic kernel
                           // keeps a mantissa plane followed by an exponent plane.
    int64_t slot_stride;
    int32_t slot_base;     // first scratch region of this launch (concurrent launches own disjoint regions)
    int32_t *px;  // sparse posterior output
    int32_t *py;
    float *pp;
    float threshold;
    int32_t wcap;  // ring capacity (cells per anti-diagonal) of the generic kernel
    float *Bv;     // dense dump of the backward match state (debug launches only)
    int32_t *Be;
    float *ring;   // generic kernel, global-ring variant: 18*wcap floats per resident wave (bands too wide for LDS)
    // Baum-Welch E-step launches only (k_dp_generic<.., EM = true>)
    float *Fx;       // forward sx, sy, lx, ly planes: 4 * slot_stride floats per resident wave
    double *em_T;    // [NPR_MAX_MODELS][25] expected transition counts, accumulated with atomics
    double *em_E;    // [NPR_MAX_MODELS][EM_BINS] expected emission counts (compact bins, see EM_BINS)
};

// compact emission bins of the E-step: 16 match [x*4+y], 4 shortGapX [x], 4 longGapX [x], 4 shortGapY [y], 4 longGapY [y]
constexpr int EM_BINS = 32;

struct CompactArgs {
    const Task *tasks;
    const TaskOut *outs;
    const int64_t *dst_off;
    int32_t ntasks;
    const int32_t *px;
    const int32_t *py;
    const float *pp;
    int32_t *cx;
    int32_t *cy;
    float *cp;
};

// launchers (npr_kernels.hip)
int launch_generic(const KernelArgs &a, int grid, int threads, size_t lds_bytes, bool dense, bool global_ring, void *stream);
int launch_em(const KernelArgs &a, int grid, size_t lds_bytes, bool global_ring, void *stream);
size_t em_extra_lds_bytes();
int launch_compact(const CompactArgs &a, void *stream);
int launch_stair(const KernelArgs &a, int R, int grid, void *stream);
int launch_wide(const KernelArgs &a, int R, int NW, int grid, void *stream);
int launch_em_stair(const KernelArgs &a, int R, int grid, void *stream);
size_t em_stair_lds_bytes();
// device MEA stage (npr_mea.hip); offsets are per read, prefix sums with n_reads + 1 entries
struct MeaArgs {
    const Task *tasks;
    const TaskOut *outs;
    int32_t ntasks, n_reads;
    const int32_t *px, *py;  // posterior pairs as the DP kernels left them (per task at Task::pair_off)
    const float *pp;
    const int64_t *rx_off;   // reference positions: lX + 1 entries per read in cnt / start
    const int64_t *ry_off;   // read positions: lY entries per read in colsum
    const int64_t *rp_off;   // pairs: the read's slice of sx / sy / sq / back
    int32_t *cnt, *start, *colsum;
    int32_t *sx, *sy, *sq, *back;
    // the kept pairs (weight above matchGamma) of every read, at the read's slice too: by id = in (x, y) order (kx, ky, kq and
    // the chain's back pointers kback), and in the order the chain visits them (vrec: read position, weight, id, 0)
    int32_t *kx, *ky, *kq, *kback;
    int4 *vrec;
    int32_t *kept;           // per read: kept pairs; -1: the ring kernel took the read (back pointers over the sorted pairs)
    // the pieces a read's chain problem is cut into (k_mea_cuts): np[r] planned pieces, boundaries pb[pboff[r] + 0 .. np[r]] among
    // the kept pairs, the last pair of each piece's heaviest chain pbest[poff[r] + j]; lane s of k_mea_chain_lanes takes piece
    // lane_piece[s] of read lane_read[s]
    const int32_t *np, *poff, *pboff, *lane_read, *lane_piece;
    int32_t *pb, *pbest;
    int32_t n_pieces;
    int32_t *best_who;       // last pair of the heaviest chain, -1 if none
    int32_t *read_flag;      // 0 or an NPR_ERR_* raised by this stage
    double gap_gamma, match_gamma;
    int32_t ring;            // entries of the prefix-maximum ring (power of two)
    const int32_t *order;    // reads by decreasing pair count: the per-read kernels take them in this order (no long read last)
    const int32_t *read_first, *read_ntasks, *task_of;  // the tasks of a read: task_of[read_first[r] + s]
    int32_t sort_lds_bytes;  // > 0: k_mea_sort_lds with this much LDS for the reads whose spans fit it
    int32_t sort_threads;    // threads of its workgroups (0: 1024)
    int32_t any_global_sort; // some read's span does not: the three global-memory kernels for those
    const int64_t *cnt_off;  // per read: its slice of cnt / start (lX + 1 entries), -1 for a read sorted in LDS
    int32_t ring_only;       // tests: every read through the LDS-ring kernel
    int32_t *ops_tmp;        // (op, length) pairs, each read's written backwards from the end of its slice
    const int64_t *ot_off;
    int32_t *n_ops, *chain_len;
    int64_t *chain_mass;
    uint32_t *ops_dense;     // one word per op: length << 2 | op
    const int64_t *od_off;
    int32_t *max_run;        // per read: its longest run (k_mea_trace) ...
    uint16_t *ops_dense16;   // ... when none of the batch exceeds 14 bits: the words' low halves, the form that crosses PCIe (else null)
};
int launch_tile(const KernelArgs &a, int R, int NW, int grid, void *stream, bool flat = false);
size_t tile_lds_bytes(int nw);
int64_t tile_scratch_cells(int64_t rows, int R);  // forward scratch (8-byte cells) of a task with that many stripe rows
// ---- device planner (npr_plan.hip): band rows, frame schedules, stripe tables and generic row offsets of a batch,
// expanded on the device from the segments' plan points (npr_band.h) ----
struct PlanSeg {
    int64_t point_first;  // first of the segment's pieces + 1 points
    int64_t band_off;     // first entry of its rows in lo / n (/ coff)
    int32_t pieces, lX, lY, pad;
};
struct SegSummary {
    int64_t cells;          // in-band lattice cells
    int64_t generic_cells;  // scratch cells of the generic kernel (rows padded to 4)
    int32_t max_width;
    int32_t bad;            // anti-diagonals without cells
    int32_t rough;          // anti-diagonals on which a band edge does not move by exactly one cell
    int32_t pad;
};
// the register classes with a frame schedule: (slots per lane, wavefronts) of kernel classes 0..6
constexpr int kSchedClasses = 7;
constexpr int kSchedR[kSchedClasses] = {1, 2, 4, 2, 2, 4, 4};
constexpr int kSchedNW[kSchedClasses] = {1, 1, 1, 4, 8, 8, 12};
struct PlanArgs {
    int32_t n_segs, fixed_mode, width;
    const PlanPoint *points;
    const PlanSeg *segs;
    int32_t *lo, *n;
    SegSummary *summary;
};
struct SchedArgs {
    int32_t n_segs;
    const PlanSeg *segs;
    const SegSummary *summary;
    const int32_t *lo, *n;
    const int64_t *ctl_off;    // per segment: first entry of its control words (pairs), -1: no candidate class
    const uint32_t *cand;      // per segment: bit c set = class c may take it
    uint32_t *ctl;
    int32_t *cls;              // out: the first candidate class whose frame can follow the band, -1: none
    int64_t *cells;            // out: forward scratch cells of that schedule
};
struct StripeArgs {
    int32_t count, R;
    const int32_t *seg_index;  // the segments that go to k_dp_tile
    const PlanSeg *segs;
    const SegSummary *summary;
    const int32_t *lo, *n;
    const int64_t *tile_off;   // per listed segment: its header in `stripes`
    Stripe *stripes;
    int64_t *rows;             // out, per listed segment
};
struct CoffArgs {
    int32_t n_segs;
    const PlanSeg *segs;
    const int32_t *n;
    uint32_t *coff;
};
int launch_plan_bands(const PlanArgs &a, void *stream);
// k_dp_stair addresses a row of a task's forward scratch as (descriptor of the task's region - row_bias) + a 32-bit byte
// offset of where lane 0 of the row would land; the bias keeps that offset non-negative for rows whose first lane is not
// lane 0 (8 * R * 63 bytes at most).  A task therefore needs 8 * cells + bias < 2^32: see stair_fits().
template <int R>
NPR_HD constexpr uint32_t row_bias() { return R == 4 ? 2048u : 1024u; }
NPR_HD constexpr bool stair_fits(int64_t rows, int slots) { return rows * slots < (int64_t(1) << 29) - 512; }
struct RowMaskArgs {
    int32_t count;              // stripe-kernel tasks
    const int32_t *seg_index;   // their segments
    const PlanSeg *segs;
    const int32_t *lo, *n;
    const int64_t *tile_off;    // per task: header of its stripe table
    const Stripe *stripes;
    const int64_t *mask_off;    // per task: first row in `out`
    uint32_t *out;
};
int launch_plan_rowmask(const RowMaskArgs &a, void *stream);
// the frame schedules in chunks (npr_plan.hip): chunk_off = prefix sum of plan_sched_chunks_of(D) over the segments (device, n_segs + 1
// entries), `chunks` = plan_sched_chunk_bytes(n_chunks) of scratch, cur = n_segs + kSchedClasses ints of scratch, cand_union = OR of the segments' candidate masks
size_t plan_sched_chunk_bytes(int64_t n_chunks);
int64_t plan_sched_chunks_of(int64_t D);
int launch_plan_sched(const SchedArgs &a, const int64_t *chunk_off, int64_t n_chunks, void *chunks, int32_t *cur, uint32_t cand_union, void *stream);
int launch_plan_stripes(const StripeArgs &a, void *stream);
int launch_plan_coff(const CoffArgs &a, void *stream);
int launch_encode(uint8_t *seq, int64_t n, void *stream);

// k_align_stats (npr_stats.hip): per-read reductions over aligned pairs
struct StatsSeg {  // a piece of a read's window whose base codes lie in `seq`: reference [xs, xe) at x_off, read [ys, ye) at y_off
    int32_t xs, xe, ys, ye;
    int64_t x_off, y_off;
};
struct StatsArgs {
    int32_t n_reads;
    const int64_t *ops_off;  // [n_reads + 1]
    const uint32_t *ops;     // one word per cigar op: length << 2 | op
    const int32_t *seg_off;  // [n_reads + 1]
    const StatsSeg *segs;
    const uint8_t *seq;      // base codes 0..4
    int32_t *out;            // [n_reads][NPR_STATS_WORDS]
};
int launch_align_stats(const StatsArgs &a, void *stream);
constexpr float EXPECT_FIXED_ONE = 1099511627776.0f;  // 2^40
struct ExpectArgs {
    const Task *tasks;
    const TaskOut *outs;
    int32_t ntasks;
    const int32_t *px, *py;
    const float *pp;
    const uint8_t *seq;
    const uint8_t *use;     // per read: 0 = skip (NULL: all reads)
    const int64_t *target;  // per read: table row of reference position 0 of its window
    unsigned long long *expect;  // [positions][4], fixed point: units of 1 / EXPECT_FIXED_ONE
    uint8_t *seen;          // [positions]
};
int launch_base_expectations(const ExpectArgs &a, void *stream);
// NPR_MODE_RESCORE_ORIGINAL on the device (npr_stats.hip; cactus_realign --rescoreOriginalAlignment, nanopore/analyses/alignmentUncertainty.py:41):
// the mean posterior over the M columns of the guide.  k_rescore_table spreads the guide's M runs into a table of one entry per reference
// position of the read's window (the read position the guide aligns it to, -1: none); k_rescore_sum looks every posterior pair up in it where the
// DP kernels left the pairs and adds the hits in FIXED POINT: p * 2^shift is an integer for an fp32 p at or above the posterior threshold, so a
// read's sum is the exact sum of its terms whatever order the atomics land in -- and equal to npr_host.cpp's `rescore`, whose double sum over
// the sorted pair list is exact too as long as columns * 2^shift stays below 2^53.
struct RescoreArgs {
    int32_t n_reads, ntasks;
    const int64_t *run_off;  // [n_reads + 1]: the M runs of each read's guide in `runs`
    const int32_t *runs;     // (x0, y0, length) per run, window coordinates
    const int64_t *gx_off;   // [n_reads + 1]: the read's slice of `gy` (reference span of its window + 1 entries)
    int32_t *gy;
    const Task *tasks;
    const TaskOut *outs;
    const int32_t *px, *py;
    const float *pp;
    unsigned long long *sum;  // [n_reads]
    int32_t shift;
};
int launch_rescore_table(const RescoreArgs &a, void *stream);
int launch_rescore_sum(const RescoreArgs &a, void *stream);
size_t mea_chain_lds_bytes(int ring);
int launch_mea_sort(const MeaArgs &a, void *stream);
int launch_mea_chain(const MeaArgs &a, void *stream);
int launch_mea_gather(const MeaArgs &a, void *stream);
int launch_em_wide(const KernelArgs &a, int R, int NW, int grid, void *stream);
int launch_em_tile(const KernelArgs &a, int R, int grid, void *stream);  // k_em_tile: the E-step on column stripes
size_t em_tile_lds_bytes(int nw);
// k_dp_rs<R> (npr_kernel_rs.hip): the one-wavefront frame kernel in row-scaled arithmetic (npr_rs.h) -- one exponent per
// anti-diagonal row instead of one per cell.  A task's scratch region (8 bytes per cell of its frame schedule, as for
// k_dp_stair) holds the forward rows at 4 bytes per cell in its first half and the row exponents, one word per NPR_RS_K
// anti-diagonals, from byte 4 * rs_half_cells(cells) on.
#ifndef NPR_RS_K
#define NPR_RS_K 16
#endif
#ifndef NPR_RS_TOP
#define NPR_RS_TOP 85  // the renormalised maximum of a row pair lies in [2^84, 2^85)
#endif
// The certificate that one exponent per row was enough.  s = eF + eB - eTot of an anti-diagonal turns a
// forward-backward product into a posterior; the rows' maxima stay below 2^(NPR_RS_TOP + 6), so no F * 2^s or B * 2^s of that
// row exceeds 2^(NPR_RS_TOP + 6 + s), and a cell whose other factor fell below fp32's normal range (2^-126 in row units: flushed,
// or a denormal that lost bits) carries a true posterior mass below 2^(NPR_RS_TOP + 6 + s - 126 + 1).  While every s stays below
// NPR_RS_S_LIMIT that is 2^-60 per cell, and all the path mass the sweeps can have lost -- every lost path passes through a
// flushed cell on the anti-diagonal where it was flushed -- is below cells x 2^-60: nothing a posterior, a total or a cigar can
// see.  A task with a row at or above the limit (its alignment runs ~90 binary orders further below the product of the row's
// largest forward and backward values than usual: the stretch between a long deletion and a long insertion, say) reports
// TASK_RERUN instead of NPR_OK and npr_batch_run runs it again with the per-cell-exponent kernel (k_dp_stair), which has no
// such limit.  So does a task whose forward sweep arrives at the end corner with nothing: whether that band really carries
// no probability (NPR_ERR_ZERO_PROB) is for the kernel without a range limit to say.
#define NPR_RS_S_LIMIT (126 - 60 - (NPR_RS_TOP + 6) - 1)
constexpr int32_t TASK_RERUN = 1;  // TaskOut::status of such a task between the two launches (never leaves npr_batch_run)
NPR_HD constexpr int64_t rs_half_cells(int64_t cells_pad) { return (cells_pad + 63) & ~int64_t(63); }
int launch_rs(const KernelArgs &a, int R, int grid, void *stream, bool sw, bool flat);  // sw: a loaded model has short-gap switches; flat: all gap emissions are 2^-2 (npr_rs.h)
// k_dp_mid_rs (npr_kernel_mid.hip): k_dp_rs's sweeps on two wavefronts that meet in the middle -- the forward one from row 0, the backward one from
// row D, each going on past the cut against the other's stored rows.  Tasks of fewer than MID_MIN_D anti-diagonals stay with k_dp_rs.
constexpr int32_t MID_MIN_D = 4 * NPR_RS_K;
#ifndef NPR_MID_WAVES2
#define NPR_MID_WAVES2 7  // wavefronts per SIMD k_dp_mid_rs<2> is compiled for
#endif
// resident wavefronts per CU of k_dp_mid_rs<R> (80 / .. / 124 registers)
inline int mid_waves_per_cu(int R) { return R == 1 ? 24 : (R == 2 ? 4 * NPR_MID_WAVES2 : 16); }
int launch_mid_rs(const KernelArgs &a, int R, int grid, void *stream, bool sw, bool flat);
// k_dp_tile_cs (npr_kernel_tile_cs.hip): k_dp_tile's stripes in column-scaled arithmetic -- one exponent per lane (pair of lattice columns);
// sw / flat as for launch_rs
int launch_tile_cs(const KernelArgs &a, int NW, int grid, void *stream, bool sw, bool flat);
size_t tile_cs_lds_bytes(int nw);
// ... and its E-step instance (k_dp_tile_cs<.., EM>): k_em_tile's job in that arithmetic; a task whose certificate fails comes back TASK_RERUN, uncounted
int launch_em_tile_cs(const KernelArgs &a, int NW, int grid, void *stream, bool sw, bool flat);
int em_tile_cs_waves();
int em_tile_cs_waves_per_cu();
size_t rs_lds_bytes();
int em_tile_waves();
int em_tile_waves_per_cu();
size_t em_wide_lds_bytes(int nw);
size_t wide_lds_bytes(int nw);
size_t stair_lds_bytes();
size_t generic_lds_bytes(int wcap);
int generic_max_wcap();

}  // namespace npr
