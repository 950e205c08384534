/*
 * realign_oracle.h -- CPU ORACLE for the banded 5-state pair-HMM realignment path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the CPU baseline -- never as the thing measured or shipped.  The product
 * (nanopore_amd/csrc, libnprealign.so) never links, loads or calls anything in here.
 *
 * PARITY UNPINNED.  The arithmetic of the reference's hot path lives in the external C program
 * `cactus_realign` (benedictpaten/cactus `bar/`, later cPecan), invoked by system() at
 *     nanopore/analyses/utils.py:587, nanopore/analyses/alignmentUncertainty.py:41,
 *     nanopore/analyses/marginAlignSnpCaller.py:136-146.
 * submodules/cactus and submodules/sonLib are EMPTY in the snapshot (.gitmodules:31-33 gives only
 * the URL, no pinned commit), so the reference realigner can be neither compiled nor run here
 * and no reference test holds expected cigars/posteriors.  This file therefore restates the
 * published cactus/cPecan algorithm (anchors -> diagonal band -> 5-state pair-HMM forward /
 * backward in log space -> posterior match probabilities >= 0.01 -> gapGamma-reweighted maximum
 * expected accuracy chain -> cigar) in double precision and is the NORMATIVE definition for this
 * repository.  What the snapshot does pin (HMM file format + three model files, the HMM
 * post-processing known-answer test, wire formats, call-site parameter sets) is checked in
 * tests/ against fixtures.
 *
 * Conventions:
 *   X = reference (slice), Y = read.  Lattice point (x,y), 0<=x<=lX, 0<=y<=lY, means "x reference
 *   bases and y read bases consumed".  Anti-diagonal d = x+y, in-diagonal coordinate xmy = x-y.
 *   States: 0 match, 1 shortGapX (ref-only), 2 shortGapY (read-only), 3 longGapX, 4 longGapY
 *   (numbering pinned by nanopore/analyses/utils.py:617 and analyses/hmm.py:24-28).
 *   HMM file: T[from*5+to] linear probabilities, E[state*16 + x*4 + y], x = ref base, y = read
 *   base, order A,C,G,T (nanopore/mappers/blasr_hmm_0.txt; utils.py:611-619).
 *   Bases are coded 0..3 = A,C,G,T and 4 = anything else (N): flat emission.
 *   Cigar ops use SAM codes 0 = M, 1 = I (read-only), 2 = D (ref-only) (utils.py:173,602).
 */
#ifndef REALIGN_ORACLE_H
#define REALIGN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NSTATE 5
#define ORC_OP_M 0
#define ORC_OP_I 1
#define ORC_OP_D 2

#define ORC_BAND_ANCHOR 0 /* cPecan-style: anchors +- diagonalExpansion, split big rectangles */
#define ORC_BAND_FIXED 1  /* fixed width W (reference positions per read base) around the guide */

#define ORC_MODE_REALIGN 0         /* utils.py:587 */
#define ORC_MODE_RESCORE_ORIGINAL 1 /* alignmentUncertainty.py:41 */
#define ORC_MODE_ALL_POSTERIORS 2   /* marginAlignSnpCaller.py:136-146 */

typedef struct {
    int32_t band_mode;          /* ORC_BAND_* */
    int32_t diagonal_expansion; /* --diagonalExpansion (xmy units, even); utils.py:587 uses 10 */
    int32_t constraint_trim;    /* anchors dropped at both ends of every gapless guide block */
    int64_t split_threshold;    /* --splitMatrixBiggerThanThis=N: rectangles with area > N*N split */
    int32_t fixed_width;        /* W for ORC_BAND_FIXED */
    double gap_gamma;           /* --gapGamma */
    double match_gamma;         /* --matchGamma */
    double posterior_threshold; /* 0.01 */
    int32_t mode;               /* ORC_MODE_* */
} orc_params;

typedef struct {
    double T[25];
    double E[80];
} orc_hmm;

typedef struct {
    int64_t xs, ys, xe, ye; /* lattice corners of the segment (inclusive) */
    int32_t ragged_start, ragged_end;
    int64_t D;     /* number of anti-diagonals - 1 = (xe-xs)+(ye-ys) */
    int32_t *lo;   /* [D+1] first in-band xmy (segment-local coordinates), parity of d */
    int32_t *n;    /* [D+1] number of in-band cells on the diagonal (>=1) */
    int64_t *off;  /* [D+2] prefix sum of n */
    int64_t cells; /* off[D+1] */
} orc_segment;

typedef struct {
    int32_t nseg;
    orc_segment *seg;
} orc_plan;

/* ---- band / segmentation (restates SURVEY 8a rows a5.1-a5.2) ---- */
orc_plan *orc_plan_build(int64_t lX, int64_t lY, const int32_t *ops /* (op,len) pairs */, int64_t nops,
                         const orc_params *p, int32_t *status);
void orc_plan_free(orc_plan *pl);
int32_t orc_plan_nseg(const orc_plan *pl);
void orc_plan_seg_info(const orc_plan *pl, int32_t s, int64_t *info8 /* xs,ys,xe,ye,rs,re,D,cells */);
void orc_plan_seg_band(const orc_plan *pl, int32_t s, int32_t *lo, int32_t *n);

/* ---- forward / backward over one banded segment, double precision, log space ---- */
/* X,Y: base codes (0..4) of the segment (length lX, lY).  Outputs (any may be NULL):
 *   total_ll      : natural-log total probability from the forward pass
 *   total_ll_bwd  : the same quantity from the backward pass (must agree)
 *   Fm, Bm        : per in-band cell, natural-log forward / backward value of the MATCH state
 *                   (band order: diagonal major, xmy ascending); -inf for dead cells
 *   Fall, Ball    : same for all five states, [cell*5+state]
 *   pairs         : (x, y, p) triples with p >= threshold, x,y 0-based base indices in the segment,
 *                   ordered by diagonal then xmy; *npairs receives the count (capped at cap)
 * returns 0, or <0 on error (-2 total probability zero). */
int32_t orc_fb_f64(const orc_hmm *h, const uint8_t *X, int64_t lX, const uint8_t *Y, int64_t lY,
                   const int32_t *lo, const int32_t *n, int32_t ragged_start, int32_t ragged_end,
                   double threshold, double *total_ll, double *total_ll_bwd, double *Fm, double *Bm,
                   double *Fall, double *Ball, int32_t *px, int32_t *py, double *pp, int64_t cap,
                   int64_t *npairs);

/* Baum-Welch E-step over one banded segment (cactus_realign --outputExpectations, the
 * quantity cactus_expectationMaximisation sums at nanopore/analyses/utils.py:509-528): expected number of uses of
 * every transition, T_exp[from*5+to], and expected emission counts E_exp[state*16 + x*4 + y] (for the gap states the
 * count of a base is spread evenly over the four values of the other index, so that the state's marginal is the
 * count).  Cells consuming an N contribute to transitions but not to emissions.  Adds into the arrays. */
int32_t orc_expectations_f64(const orc_hmm *h, const uint8_t *X, int64_t lX, const uint8_t *Y, int64_t lY,
                             const int32_t *lo, const int32_t *n, int32_t ragged_start, int32_t ragged_end,
                             double *T_exp, double *E_exp, double *total_ll);

/* fp32 mirror of the device arithmetic (block floating point: five linear fp32 mantissas sharing one
 * int32 binary exponent per cell).  Restates DESIGN.md "device arithmetic" operation by operation so
 * that its results are bit-identical to the HIP kernels'.  Outputs as above except Fm/Bm are given
 * as (mantissa float, exponent int32) and the posterior is float. */
int32_t orc_fb_f32(const orc_hmm *h, const uint8_t *X, int64_t lX, const uint8_t *Y, int64_t lY,
                   const int32_t *lo, const int32_t *n, int32_t ragged_start, int32_t ragged_end,
                   float threshold, float *tot_m, int32_t *tot_e, float *btot_m, int32_t *btot_e,
                   float *Fm_v, int32_t *Fm_e, float *Bm_v, int32_t *Bm_e, int32_t *px, int32_t *py,
                   float *pp, int64_t cap, int64_t *npairs);

/* fp32 mirror of the device's ROW-SCALED arithmetic (realign_oracle_rs.c; nanopore_amd/csrc/npr_rs.h: the kernels of the
 * bands one wavefront's frame holds): plain fp32 cells relative to one binary exponent per anti-diagonal row, both held
 * rows renormalised after every 16th anti-diagonal.  Same signature; Fm / Bm are the match values with their row's exponent. */
int32_t orc_fb_f32_rs(const orc_hmm *h, const uint8_t *X, int64_t lX, const uint8_t *Y, int64_t lY,
                      const int32_t *lo, const int32_t *n, int32_t ragged_start, int32_t ragged_end,
                      float threshold, float *tot_m, int32_t *tot_e, float *btot_m, int32_t *btot_e,
                      float *Fm_v, int32_t *Fm_e, float *Bm_v, int32_t *Bm_e, int32_t *px, int32_t *py,
                      float *pp, int64_t cap, int64_t *npairs);

/* ---- maximum expected accuracy chain + cigar ---- */
/* pairs in any order; lX,lY the full spans; out_ops receives (op,len) pairs (capacity cap_ops pairs).
 * Returns number of ops, or <0 on error. score = mean posterior of the chosen pairs. */
int64_t orc_mea_cigar(int64_t lX, int64_t lY, const int32_t *px, const int32_t *py, const double *pp,
                      int64_t npairs, double gap_gamma, double match_gamma, int32_t *out_ops,
                      int64_t cap_ops, double *score, int32_t brute_force);

/* mean posterior over the M columns of the guide */
double orc_rescore(const int32_t *ops, int64_t nops, const int32_t *px, const int32_t *py,
                   const double *pp, int64_t npairs);

/* ---- whole read: guide -> plan -> F/B per segment -> pairs -> (MEA cigar | rescore | dump) ---- */
typedef struct {
    int32_t status;
    int64_t cells;
    double total_ll; /* sum over segments */
    double score;
    int64_t nops;
    int64_t npairs;
} orc_read_result;

/* precision: 0 = fp64 log space, 1 = fp32 mirror.
 * out_ops capacity cap_ops pairs; pairs capacity cap_pairs (coordinates are absolute in X / Y). */
int32_t orc_realign_read(const orc_hmm *h, const orc_params *p, int32_t precision, const uint8_t *X,
                         int64_t lX, const uint8_t *Y, int64_t lY, const int32_t *guide_ops,
                         int64_t n_guide_ops, int32_t *out_ops, int64_t cap_ops, int32_t *px, int32_t *py,
                         double *pp, int64_t cap_pairs, orc_read_result *res);

/* batch of reads (CSR layout), OpenMP over reads; used by bench.py's cpu_baseline leg.
 * Only totals are returned: per-read cells, score, status. threads<=0 -> omp default. */
/* orc_realign_read with the fp32 arithmetic named per segment (matrix split), as the product reports it
 * (npr_batch_segment_arith): seg_arith[s] 0 = per-cell exponents (orc_fb_f32), 1 = row-scaled (orc_fb_f32_rs); NULL or
 * fewer than the read's segments: 0.  Only read when precision == 1. */
int32_t orc_realign_read_arith(const orc_hmm *h, const orc_params *p, int32_t precision, const int32_t *seg_arith,
                               int64_t n_seg_arith, const uint8_t *X, int64_t lX, const uint8_t *Y, int64_t lY,
                               const int32_t *guide_ops, int64_t n_guide_ops, int32_t *out_ops, int64_t cap_ops,
                               int32_t *px, int32_t *py, double *pp, int64_t cap_pairs, orc_read_result *res);
/* ... and for a batch: seg_off[nreads + 1] into seg_arith (NULL: all 0) */
int32_t orc_realign_batch_arith(const orc_hmm *h, const orc_params *p, int32_t precision, const int64_t *seg_off,
                                const int32_t *seg_arith, int64_t nreads, const uint8_t *X, const int64_t *x_off,
                                const uint8_t *Y, const int64_t *y_off, const int32_t *guide_ops, const int64_t *g_off,
                                int32_t *out_ops, const int64_t *o_off, int64_t *out_nops, double *out_score,
                                double *out_ll, int64_t *out_cells, int32_t *out_status, int32_t threads);
int32_t orc_realign_batch(const orc_hmm *h, const orc_params *p, int32_t precision, int64_t nreads,
                          const uint8_t *X, const int64_t *x_off, const uint8_t *Y, const int64_t *y_off,
                          const int32_t *guide_ops, const int64_t *g_off /* in op pairs */,
                          int32_t *out_ops, const int64_t *o_off /* capacity per read, op pairs */,
                          int64_t *out_nops, double *out_score, double *out_ll, int64_t *out_cells,
                          int32_t *out_status, int32_t threads);

/* log-add used by the fp64 passes: 0 exact (normative), 1 cPecan's piecewise-cubic approximation [RECALLED]: a global
 * switch for measuring what the reference's own approximation could move; set it back to 0 afterwards. */
void orc_set_logadd_kind(int32_t kind);
int32_t orc_get_logadd_kind(void);
double orc_logadd(double a, double b);

int32_t orc_version(void);

#ifdef __cplusplus
}
#endif
#endif
