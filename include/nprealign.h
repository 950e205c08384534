/*
 * nprealign.h -- C ABI of libnprealign.so: batched banded five-state pair-HMM realignment of
 * nanopore reads on MI355X (gfx950), the drop-in for the reference's per-read `cactus_realign`
 * subprocess.
 *
 * What it replaces.  The reference has no FFI for this path: it has a PROCESS boundary.  One
 * `cactus_realign` process is forked per SAM record by sonLib's system() at
 *     nanopore/analyses/utils.py:587            (realign; fan-out loop utils.py:565-570,
 *                                                gather/splice utils.py:591-609)
 *     nanopore/analyses/alignmentUncertainty.py:41        (rescore the original alignment)
 *     nanopore/analyses/marginAlignSnpCaller.py:136-146   (dump all posterior match probabilities)
 * with inputs "reference FASTA, read FASTA, exonerate cigar on stdin, --diagonalExpansion,
 * --splitMatrixBiggerThanThis, --gapGamma, --matchGamma, --loadHmm" and outputs "one cigar line
 * on stdout (+ score), optional `refPos readPos prob` TSV".  Each entry point below cites the
 * piece of that contract it stands for.  INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Conventions: plain pointers and sizes only; the caller owns every input buffer (borrowed for the
 * duration of the call); every function returns NPR_OK (0) or a negative NPR_ERR_* code and, where an
 * `err` buffer is given, writes a message into it; no exceptions, no globals, no stdout.  One
 * npr_ctx per (thread, device); calls on one ctx must be serialised by the caller.  There is NO CPU
 * fallback: npr_create fails with NPR_ERR_NO_DEVICE when no gfx950 GPU / HIP runtime is usable.
 *
 * Coordinates: X = reference, Y = read; cigar ops are (op,len) int32 pairs with SAM op codes
 * 0 = M, 1 = I (read only), 2 = D (reference only) -- the mapping realignSamFile3TargetFn relies on
 * (utils.py:602) and getExonerateCigarFormatString emits (utils.py:173).
 */
#ifndef NPREALIGN_H
#define NPREALIGN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPR_ABI_VERSION 1

/* error codes */
#define NPR_OK 0
#define NPR_ERR_INVALID (-1)       /* bad argument / guide cigar is not a global alignment (utils.py:381-382) */
#define NPR_ERR_ZERO_PROB (-2)     /* total probability of the banded model is zero */
#define NPR_ERR_CAPACITY (-3)      /* an output buffer (ops / posterior pairs) was too small */
#define NPR_ERR_MODEL (-4)         /* HMM has a transition outside the five-state cell update */
#define NPR_ERR_NO_DEVICE (-5)     /* no usable gfx950 device: there is no CPU fallback */
#define NPR_ERR_HIP (-6)           /* HIP runtime error (message in err / npr_last_error) */
#define NPR_ERR_BAND_TOO_WIDE (-7) /* an anti-diagonal has more in-band cells than the kernels support */
#define NPR_ERR_NOMEM (-8)
#define NPR_ERR_STATE (-9)         /* call sequence violated (e.g. finish before run) */

/* cigar op codes (SAM numbering) */
#define NPR_OP_M 0
#define NPR_OP_I 1
#define NPR_OP_D 2

/* band construction */
#define NPR_BAND_ANCHOR 0 /* cactus_realign's: anchors from the guide's M columns +- diagonalExpansion,
                             rectangles between distant anchors, split above splitMatrixBiggerThanThis^2 */
#define NPR_BAND_FIXED 1  /* fixed width W around the guide path (BASELINE.json "band=100/200") */

/* what to return per read: the three call sites of cactus_realign */
#define NPR_MODE_REALIGN 0          /* utils.py:587: new MEA cigar + score */
#define NPR_MODE_RESCORE_ORIGINAL 1 /* alignmentUncertainty.py:41: --rescoreOriginalAlignment
                                       --rescoreByPosteriorProbIgnoringGaps: guide ops kept, score = mean
                                       posterior of its M columns */
#define NPR_MODE_ALL_POSTERIORS 2   /* marginAlignSnpCaller.py:136-146: --outputAllPosteriorProbs; the MEA
                                       cigar is produced as well (stdout of that call) */
#define NPR_MODE_EXPECTATIONS 3     /* utils.py:509-528: the batch is staged for npr_batch_expectations (the E-step of
                                       cactus_expectationMaximisation); npr_batch_run / finish behave as NPR_MODE_REALIGN.
                                       A batch staged in another mode may be laid out for realignment only (scratch regions
                                       of their own size, long reads on two wavefronts): npr_batch_expectations then
                                       returns NPR_ERR_STATE */

#define NPR_MAX_MODELS 8

typedef struct npr_ctx npr_ctx;
typedef struct npr_batch npr_batch;
typedef struct npr_plan npr_plan;

/* The command-line options of cactus_realign used by the reference's three call strings. */
typedef struct {
    int32_t band_mode;           /* NPR_BAND_* */
    int32_t diagonal_expansion;  /* --diagonalExpansion=10 (utils.py:587) */
    int32_t constraint_trim;     /* anchors trimmed at both ends of each gapless block (cPecan default 14) */
    int64_t split_threshold;     /* --splitMatrixBiggerThanThis: 3000 realign (utils.py:587), 100 analyses
                                    (alignmentUncertainty.py:41, marginAlignSnpCaller.py:136) */
    int32_t fixed_width;         /* W for NPR_BAND_FIXED, >= 2 (narrower: NPR_ERR_INVALID for the read) */
    double gap_gamma;            /* --gapGamma  (abstractMapper.py:25 default 0.5) */
    double match_gamma;          /* --matchGamma (abstractMapper.py:25 default 0.0) */
    double posterior_threshold;  /* 0.01 */
    int32_t mode;                /* NPR_MODE_* */
    int32_t max_pairs_per_base;  /* capacity of the sparse posterior list per read base (0 -> 6) */
} npr_params;

typedef struct {
    int32_t status;    /* NPR_OK or NPR_ERR_* for this read only: one bad read does not fail the batch */
    int32_t n_segments;
    int64_t cells;     /* in-band lattice cells processed (forward + backward + posterior test each) */
    double loglik;     /* natural-log total probability summed over segments */
    double loglik_bwd; /* same from the backward pass (consistency check) */
    double score;      /* REALIGN / ALL_POSTERIORS: mean posterior of the MEA pairs; RESCORE: mean posterior of
                          the guide's M columns -- what the reference reads back as pA.score
                          (alignmentUncertainty.py:48) */
    int64_t n_ops;     /* number of (op,len) pairs of the output cigar */
    int64_t n_pairs;   /* number of posterior pairs >= threshold */
} npr_read_result;

typedef struct {
    int64_t n_reads, n_tasks;
    int64_t cells;          /* total in-band cells of the batch */
    int64_t diagonals;      /* total anti-diagonals */
    int64_t max_width;      /* widest anti-diagonal (cells) */
    int64_t device_bytes;   /* device memory held by the batch */
    int64_t slots;          /* resident wavefront slots used by the DP launch */
    int32_t kernel_variant; /* kernel that carries most cells: 0 = generic LDS-ring kernel, 1 = register kernel on a frame
                               that follows the anti-diagonal (k_dp_stair / k_dp_wide), 2 = register kernel on column
                               stripes (k_dp_tile) */
} npr_batch_stats;

/* ---- library / context ---- */
int32_t npr_abi_version(void);
const char *npr_strerror(int32_t code);

/* device_id >= 0.  Fails with NPR_ERR_NO_DEVICE if HIP or the device is unavailable. */
int32_t npr_create(int32_t device_id, npr_ctx **out, char *err, size_t errlen);
void npr_destroy(npr_ctx *ctx);
const char *npr_last_error(npr_ctx *ctx);
/* Context options.  NPR_OPT_OVERLAP (value 0 / 1 / 2, default 0): the context is one of several on its device whose batches are
 * in flight together -- a pipelined job stages batch k+1 and finishes batch k-1 while batch k is in its DP pass
 * (nanopore_amd/job.py; the reference's analogue is jobTree running several cactus_realign processes at once,
 * /root/reference/Makefile:1 maxThreads).  The contexts of a device share its forward scratch and take turns in it; with this
 * option the device MEA stage keeps its tables in buffers of the context's own instead (so npr_batch_finish need not wait for
 * another batch's DP pass), and the DP launches of narrow bands take four wavefronts per SIMD instead of seven, so that the staging
 * and MEA kernels of the other batches find slots and registers beside them (a persistent launch that fills the chip leaves room for
 * nothing; nanopore_amd/job.py's default since the end of round 5).  Value 2: the tables of its own only -- the next batch's DP pass
 * starts when it is staged, not when this batch's MEA stage has given the shared scratch back; the MEA kernels take the slots the
 * DP pass leaves as its wavefronts run out.  Results do not change. */
#define NPR_OPT_OVERLAP 1
/* NPR_OPT_RELEASE_SCRATCH (an action; value 2: only the context's cache of released device buffers, the scratch stays): the device's forward scratch (shared by the contexts of the device; the
 * next batch that needs it allocates it again) and this context's cache of released device buffers go back to the driver.  For
 * a process that stays alive after a big batch (the parent of a pipeline, a test session) next to others that need the HBM. */
#define NPR_OPT_RELEASE_SCRATCH 2
/* TEST AND BRING-UP SWITCHES (all 0 by default; none changes a result -- every kernel class computes the same bits or is checked
 * against the same oracle -- only which kernel runs or where a table lives).  They select the code paths the parity tests
 * compare with each other; a caller of the drop-in path never sets them.  Since round 4 these are context options, not
 * environment variables: what the library runs does not depend on the caller's environment.  (Still read from the environment,
 * and changing no choice of kernel: NPR_TIMING=1 stage times on stderr, NPR_POISON=<byte> device buffers filled when handed
 * out, NPR_TILE_PROF=1 wait cycles of the stripe kernel, NPR_HOST_THREADS=<n> host worker threads.) */
#define NPR_OPT_KERNEL 3           /* 1: the any-band kernel (k_dp_generic) for every task */
#define NPR_OPT_ARITH 4            /* 1: one exponent per cell (k_dp_stair) instead of one per row (k_dp_rs / k_dp_mid_rs) */
#define NPR_OPT_PAIR 5             /* a read's two sweeps on two wavefronts: 0 the default rule (row-scaled arithmetic: every task of 64+ anti-diagonals), 1 never, 2 the tasks longer than a fair share, 3 always */
#define NPR_OPT_NO_TILE 6          /* 1: no stripe kernel (wide bands take k_dp_wide / k_dp_generic) */
#define NPR_OPT_NO_WIDE 7          /* 1: no multi-wavefront frame kernel */
#define NPR_OPT_TILE_RS 8          /* the stripe kernel: 0 / 1 column-scaled arithmetic (k_dp_tile_cs, the default), 2 one exponent per cell (k_dp_tile) */
#define NPR_OPT_TILE_WAVES 9       /* wavefronts per stripe task (1 .. 8; 0: default 4) */
#define NPR_OPT_WAVES_PER_CU 10    /* resident wavefronts / workgroups per CU of every DP launch (0: per class) */
#define NPR_OPT_CLASS_MIN 11       /* smallest frame class considered */
#define NPR_OPT_VARIABLE_SCRATCH 12 /* scratch regions sized per task: 0 above 32 GB, 1 always, 2 never */
#define NPR_OPT_HOST_MEA 13        /* 1: chain + cigar on the host in realign mode too */
#define NPR_OPT_MEA_RING_ONLY 14   /* 1: every read through the LDS-ring chain kernel */
#define NPR_OPT_MEA_GLOBAL_SORT 15 /* 1: the sort's tables in HBM whatever the span */
#define NPR_OPT_MEA_OWN_SCRATCH 16 /* 1: the MEA tables in buffers of the context's own, not in the forward scratch */
#define NPR_OPT_EM_GENERIC 17      /* 1: the E-step on the any-band kernel */
#define NPR_OPT_EM_SERIAL 18       /* 1: the E-step's launches one after the other */
#define NPR_OPT_EM_WAVES 19        /* wavefronts per CU of k_em_stair (0: per class) */
#define NPR_OPT_MEA_WIDE_OPS 20    /* 1: the packed cigars cross PCIe as whole words even when every run fits 14 bits */
#define NPR_OPT_EM_TILE 21         /* the E-step of stripe tasks: 0 column-scaled arithmetic first (k_dp_tile_cs's E-step instance; what its certificate refuses goes to k_em_tile), 1 k_em_tile only, 2 (tests) as 0 with every other task refused */
#define NPR_OPT_COUNT 22
int32_t npr_ctx_option(npr_ctx *ctx, int32_t option, int64_t value);

/* --loadHmm=<file> (utils.py:586-587): the 25 transition and 80 emission PROBABILITIES exactly as they
 * stand in the two-line model file (nanopore/mappers/blasr_hmm_0.txt).  slot in [0, NPR_MAX_MODELS): reads
 * pick a slot through `model_slot` (per-read-type models, scripts/modifyHmm.py).  T == NULL installs the
 * stock model used when the reference passes no --loadHmm (abstractMapper.py:36-37). */
int32_t npr_set_hmm(npr_ctx *ctx, int32_t slot, const double *T25, const double *E80);

/* ---- batch: replaces the per-read fan-out / gather of utils.py:557-609 ---- */
/* Stage 1 (host + H2D): band construction, packing, upload.  Sequences are ASCII (ACGT, any case; anything
 * else is N).  The reference side is a table of n_refs sequences, ref[ref_off[k] .. ref_off[k+1]) -- the
 * contigs of the reference FASTA (argv[1] of cactus_realign) -- and read i is aligned against sequence
 * ref_index[i] (the cigar's target name, utils.py:570); ref_index == NULL means n_refs == n_reads and read i
 * uses sequence i (per-read slices).  read i = read[read_off[i] .. read_off[i+1]) (aR.query, utils.py:570);
 * guide i = (op,len) pairs guide_ops[2*guide_off[i] .. 2*guide_off[i+1]) and must be GLOBAL over both
 * sequences (the chained records of utils.py:313-386).  Only the parts of a reference sequence that a read's
 * band touches are uploaded, so a 4.6 Mb contig shared by 50 k reads costs nothing extra.
 * model_slot may be NULL (all 0). */
int32_t npr_batch_create(npr_ctx *ctx, const npr_params *params, int64_t n_reads, int64_t n_refs,
                         const uint8_t *ref, const int64_t *ref_off, const int32_t *ref_index,
                         const uint8_t *read, const int64_t *read_off, const int32_t *guide_ops,
                         const int64_t *guide_off, const int32_t *model_slot, npr_batch **out);
/* The same with guides that carry coordinates, like the exonerate cigar line the reference pipes into cactus_realign
 * (`cigar: query qstart qend + target tstart tend + score ops`, getExonerateCigarFormatString at
 * nanopore/analyses/utils.py:173-186, consumed at utils.py:587): guide i starts at reference position
 * guide_start[2*i] and read position guide_start[2*i+1] (0-based) and spans what its ops consume; the realignment is
 * confined to that window, exactly as cactus_realign realigns only the sub-sequences a cigar covers.  Output cigars
 * cover the window; posterior coordinates stay absolute in the sequences.  guide_start == NULL: every guide starts
 * at (0, 0) and must be global, i.e. npr_batch_create. */
int32_t npr_batch_create_at(npr_ctx *ctx, const npr_params *params, int64_t n_reads, int64_t n_refs,
                            const uint8_t *ref, const int64_t *ref_off, const int32_t *ref_index,
                            const uint8_t *read, const int64_t *read_off, const int32_t *guide_ops,
                            const int64_t *guide_off, const int64_t *guide_start, const int32_t *model_slot,
                            npr_batch **out);
/* Stage 2 (device): forward + backward + posterior extraction for every read of the batch; inputs are
 * resident in HBM.  Blocks until done; kernel_ms (nullable) receives the HIP-event time of the DP launch
 * measured on the context's stream.  May be called repeatedly (benchmarks). */
int32_t npr_batch_run(npr_batch *b, float *kernel_ms);
/* Stage 3: MEA chain / rescore, cigars.  In NPR_MODE_REALIGN and NPR_MODE_ALL_POSTERIORS the chain and the cigar are computed on
 * the device from the posterior pairs where they lie (integer arithmetic: the same ops and scores as the host stage, npr_mea_cigar)
 * and only the run-length ops come back.  In NPR_MODE_RESCORE_ORIGINAL the guide's M columns -- spread into a table on the device
 * when the batch was staged -- are looked up where the pairs lie and summed in fixed point (the host stage's double exactly, npr_rescore);
 * eight bytes per read come back, and the cigars are the guide's (made when npr_batch_ops / npr_batch_ops_packed ask).  No mode moves
 * the pairs over PCIe unless npr_batch_pairs asks for them.  Batches whose tables do not fit the device, a rescore sum that could not be
 * exact (a threshold below 2^-20, a guide of 2^23 M columns) and NPR_OPT_HOST_MEA copy the pairs to the host and finish there. */
int32_t npr_batch_finish(npr_batch *b);
void npr_batch_destroy(npr_batch *b);

int32_t npr_batch_get_stats(const npr_batch *b, npr_batch_stats *st);
/* Diagnostics: how the batch's DP problems (segments) were spread over the kernel classes.  tasks[c] / cells[c] for
 * class c (either may be NULL), capacity `cap` entries; returns the number of classes (19):
 *   0-2  register kernel, one wavefront per task, 64 / 128 / 256 slots (k_dp_stair<1|2|4>)
 *   3-6  register kernel, 4 / 8 / 8 / 12 wavefronts per task, 512 / 1024 / 2048 / 3072 slots (k_dp_wide)
 *   7-9  generic kernel, LDS ring for at most 512 / 1024 / 2270 cells per anti-diagonal;  10  generic kernel, HBM ring
 *   11   register kernel on column stripes, any width (k_dp_tile; k_em_tile for npr_batch_expectations); takes what 3-10
 *        would take unless NPR_OPT_NO_TILE is set
 *   12-14  classes 15-17 with the forward and the backward sweep of a task on two wavefronts at once, k_dp_mid_rs<1|2|4> (round 5): the
 *        sweeps start at the two ends and MEET IN THE MIDDLE, each going on past the cut against the rows the other one stored -- a task's
 *        serial chain halves, no more bytes or instructions than k_dp_rs, the same bits; the default for every row-scaled task of 64+
 *        anti-diagonals (shorter ones stay in 15-17).  NPR_OPT_PAIR 1: never, 2: only the tasks longer than a wavefront's fair share of
 *        their class as far as second wavefronts are free
 *   15-17  classes 0-2 in row-scaled arithmetic (k_dp_rs<1|2|4>: one exponent per anti-diagonal row of the wavefront instead of
 *        one per cell, about 1.4 times the cells per second): every task of 0-2 unless NPR_OPT_ARITH = 1 is set or a loaded model's
 *        values can grow from one anti-diagonal to the next.  A task for which one exponent per row was not enough (a stretch of
 *        its alignment ~110 binary orders below the row's largest values: an indel of 70+ bases) is run again by npr_batch_run
 *        with the kernel of 0-2; npr_batch_segment_arith says which arithmetic a segment's results come from.
 *   18   class 11's column stripes in column-scaled arithmetic (k_dp_tile_cs: one exponent per lane of a stripe; the default for the stripe
 *        tasks since round 6, same bits as class 11; a task without its per-lane range certificate runs again in class 11's kernel) */
int32_t npr_batch_class_stats(const npr_batch *b, int64_t *tasks, int64_t *cells, int32_t cap);
/* Which device arithmetic each segment (matrix split) of each read ran in, in read order: seg_off[n_reads + 1], arith[seg_off
 * [n_reads]] (pass arith == NULL for the offsets alone).  0: one exponent per cell (npr_cell.h: k_dp_tile, k_dp_generic, k_dp_wide
 * k_dp_stair); 1: one exponent per anti-diagonal row (npr_rs.h: k_dp_rs, k_dp_mid_rs; classes 15-17, 12-14).  Both are fp32 evaluations of the same recurrences (cactus_realign's forward / backward pass, reference call
 * site nanopore/analyses/utils.py:587) within the stated 1e-4 of the fp64 oracle; the parity tests ask so that they can
 * compare bit for bit with the matching CPU restatement (oracle/realign_oracle_f32.c / realign_oracle_rs.c). */
int32_t npr_batch_segment_arith(const npr_batch *b, int64_t *seg_off, int32_t *arith, int64_t cap);
/* results, valid after npr_batch_finish */
int32_t npr_batch_results(const npr_batch *b, npr_read_result *out /* [n_reads] */);
/* output cigars, CSR: ops_off[n_reads+1] (in op pairs), ops[2*ops_off[n_reads]].  Pass ops == NULL to get
 * only the offsets (two-pass sizing). */
int32_t npr_batch_ops(const npr_batch *b, int64_t *ops_off, int32_t *ops, int64_t cap_pairs);
/* the same cigars, one 32-bit word per operation: length << 2 | op (the form the device MEA stage produces and a sharded
 * job ships between ranks); words[ops_off[n_reads]].  Pass words == NULL for the offsets only. */
int32_t npr_batch_ops_packed(const npr_batch *b, int64_t *ops_off, uint32_t *words, int64_t cap_words);
/* sparse posteriors (>= threshold), CSR by read, sorted by (x, y); x is the 0-based reference coordinate in
 * the read's slice, y the 0-based read coordinate: the `refPos readPos prob` TSV of
 * --outputAllPosteriorProbs (marginAlignSnpCaller.py:149).  After a device-side npr_batch_finish the pairs are
 * still in HBM: the first call with x != NULL copies and sorts them (pair_off alone costs nothing). */
int32_t npr_batch_pairs(const npr_batch *b, int64_t *pair_off, int32_t *x, int32_t *y, float *p, int64_t cap);
/* TEST HOOK (tests/test_gpu_parity.py: the finish stages against malformed input).  Replaces, on the device, the posterior pairs the DP pass left for
 * `read` -- which must have a single segment -- by the n given ones (window coordinates; n at most the list's capacity), as if the DP kernel had
 * written them with status `task_status` (NPR_OK, or e.g. NPR_ERR_CAPACITY for a list that overflowed).  Between npr_batch_run and npr_batch_finish.
 * Nothing in the product calls it. */
int32_t npr_batch_debug_set_pairs(npr_batch *b, int64_t read, const int32_t *x, const int32_t *y, const float *p, int64_t n, int32_t task_status);

/* ---- post-alignment statistics on the device ----
 * The per-read integer reductions the reference's analyses make by walking every aligned pair in Python:
 * nanopore/analyses/coverage.py:10-95 (ReadAlignmentCoverageCounter), substitutions.py:9-56, indels.py:9-45.
 * One record of NPR_STATS_WORDS int32 per read:
 *   [0] matches (same base, reference base in ACGT)   [1] mismatches (both in ACGT, different)   [2] aligned pairs against N
 *   [3] aligned pairs (M columns)
 *   [4] read insertions between aligned pairs, [5] their total length; [6] read deletions, [7] their total length
 *       (coverage.py:36-41 / indels.py:22-25: runs between two consecutive aligned pairs)
 *   [8] read bases / [9] reference bases the cigar consumes before its first aligned pair, [10] / [11] after its last one
 *       (what a global alignment adds, coverage.py:42-58)
 *   [12] reference bases, [13] read bases the cigar consumes; [14] 0 or NPR_ERR_INVALID (cigar runs out of its sequences)
 *   [15 + 5 * r + q] aligned pairs of reference base r against read base q, A C G T N (substitutions.py:13-19)
 * npr_batch_align_stats: the alignments npr_batch_finish just produced, where they lie (packed cigars and base codes are
 * still in HBM after the device MEA stage; otherwise the cigars are uploaded).  npr_align_stats: any alignments, e.g. the
 * records of a mapper's SAM file: read i = read[read_off[i] ..) against reference sequence ref_index[i] (NULL: i), cigar
 * i = (op, length) pairs ops[2 * ops_off[i] ..) starting at reference position start[2 * i] and read position
 * start[2 * i + 1] (start == NULL: 0, 0).  Sequences are ASCII as in npr_batch_create. */
#define NPR_STATS_WORDS 40
int32_t npr_batch_align_stats(npr_batch *b, int32_t *stats /* [n_reads][NPR_STATS_WORDS] */);
int32_t npr_align_stats(npr_ctx *ctx, int64_t n_reads, int64_t n_refs, const uint8_t *ref, const int64_t *ref_off,
                        const int32_t *ref_index, const uint8_t *read, const int64_t *read_off, const int32_t *ops,
                        const int64_t *ops_off, const int64_t *start, int32_t *stats /* [n_reads][NPR_STATS_WORDS] */);

/* Expected base counts per reference position from the posterior pairs of a finished batch, on the device (SURVEY.md 8f
 * next #4): what marginAlignSnpCaller.py:150-155 collates from the --outputAllPosteriorProbs files, one text line at a time:
 * every pair (refPos, readPos, p) of a selected read adds p to expect[(first row of its reference + refPos) * 4 + base] for
 * the read's base at readPos (A C G T; other bases add nothing) and sets seen[...].  use[i] != 0 selects read i (NULL: all) --
 * the caller's coverage sampling; the reference table is n_refs sequences of ref_len[k] positions, rows in that order.
 * Accumulated in 64-bit fixed point (units of 2^-40): a sum is the exact sum of its fp32 terms whatever the order, so the
 * table is the same from run to run and equals a sequential double-precision sum of the same pairs. */
int32_t npr_batch_base_expectations(npr_batch *b, const uint8_t *use, int64_t n_refs, const int64_t *ref_len, double *expect,
                                    uint8_t *seen);

/* Baum-Welch E-step over the staged batch with the models currently installed: what
 * `cactus_realign --outputExpectations` produces per alignment and cactus_expectationMaximisation sums over all of
 * them in every EM iteration (nanopore/analyses/utils.py:509-528).  T_exp[slot*25 + from*5 + to] and
 * E_exp[slot*80 + state*16 + x*4 + y] receive the expected transition / emission counts of the reads that use model
 * `slot` (gap states: a base's count is spread evenly over the other index), loglik[slot] the summed natural-log
 * likelihood.  Arrays are NPR_MAX_MODELS slots long and are overwritten.  The band / split plan of the batch is
 * unaffected, so the call can be repeated after npr_set_hmm for the next iteration. */
int32_t npr_batch_expectations(npr_batch *b, double *T_exp, double *E_exp, double *loglik, float *kernel_ms);

/* debugging / parity aid: dense per-cell match-state forward and backward values of read i, task-major,
 * band order, as (mantissa, exponent) block-floating-point pairs (value = mant * 2^exp).  Runs the read
 * again on the device.  Buffers sized npr_read_result.cells. */
int32_t npr_batch_dense(npr_batch *b, int64_t read_index, float *Fm_v, int32_t *Fm_e, float *Bm_v,
                        int32_t *Bm_e, int64_t cap);

/* debugging / parity aid for the north-star kernel: the forward match values k_dp_rs (row-scaled arithmetic: one exponent per
 * anti-diagonal row, renormalised every 16 rows) leaves in its scratch for the backward sweep, read i, task-major, band order,
 * as (value, exponent of the cell's row): F = value * 2^exponent.  A cell more than ~211 binary orders below its rows' maximum
 * is 0 (flushed) -- the range certificate (npr_batch_segment_arith reports the tasks that ran again without it) bounds the
 * posterior mass such cells can carry by 2^-60.  Runs the read's tasks again, one at a time; NPR_ERR_STATE when a segment of
 * the read is not in a class k_dp_rs runs.  Buffers sized npr_read_result.cells. */
int32_t npr_batch_rs_forward(npr_batch *b, int64_t read_index, float *Fm_v, int32_t *Fm_e, int64_t cap);

/* Test aid: npr_batch_create expands band rows, frame schedules, stripe tables and row offsets on the device; this
 * recomputes every task of the batch with the host planner (the npr_plan_* functions below, the ones the tests pin
 * against the oracle) and compares entry by entry.  Returns the number of tasks that differ (0 = identical) or NPR_ERR_*. */
int64_t npr_batch_plan_check(npr_batch *b, const int32_t *guide_ops /* as given to npr_batch_create: a batch keeps no copy */);

/* One call = create + run + finish + copy-out + destroy, for callers that do not need staging. */
int32_t npr_realign_batch(npr_ctx *ctx, const npr_params *params, int64_t n_reads, int64_t n_refs,
                          const uint8_t *ref, const int64_t *ref_off, const int32_t *ref_index,
                          const uint8_t *read, const int64_t *read_off, const int32_t *guide_ops,
                          const int64_t *guide_off, const int32_t *model_slot, npr_read_result *results,
                          int64_t *ops_off, int32_t *ops, int64_t cap_op_pairs);

/* ---- host logic, callable without a GPU (unit tests of the boundary) ---- */
/* band / segmentation of one read (cactus_realign stages a5.1-a5.2 of SURVEY.md 8a) */
int32_t npr_plan_create(const npr_params *params, int64_t lX, int64_t lY, const int32_t *guide_ops,
                        int64_t n_guide_ops, npr_plan **out);
void npr_plan_destroy(npr_plan *pl);
int32_t npr_plan_segments(const npr_plan *pl);
/* info8 = xs, ys, xe, ye, ragged_start, ragged_end, D, cells */
int32_t npr_plan_segment_info(const npr_plan *pl, int32_t seg, int64_t *info8);
/* lattice-clipped band: lo[D+1], n[D+1] */
int32_t npr_plan_segment_band(const npr_plan *pl, int32_t seg, int32_t *lo, int32_t *n);
/* Frame schedule of the register kernels for one segment (host logic, for inspection and tests): the kernels hold a
 * frame of `slots` lattice points of the current anti-diagonal, slot j = (x0 + j, y0 - j), which takes an X-step
 * (x0 += 1) into every odd anti-diagonal and a Y-step (y0 += 1) into every even one and may be rebased by one slot
 * before a step.  Per anti-diagonal d (arrays of D+1): jlo[d] = slot of the first band cell, rebase[d] in {-1,0,+1} =
 * rebase applied before the step into d (+1, towards higher x-y, only before X-steps; -1 only before Y-steps),
 * row_off[d] = offset (cells) of its row in the forward scratch, rows holding whole lanes of `slots_per_lane` slots;
 * *cells = scratch cells of the segment.  slots = 64 * slots_per_lane for k_dp_stair (slots_per_lane 1, 2, 4);
 * 512, 1024 (slots_per_lane 2), 2048 or 3072 (4) for k_dp_wide.  NPR_ERR_BAND_TOO_WIDE when the band cannot be
 * followed with that frame. */
int32_t npr_plan_frame_schedule(const npr_plan *pl, int32_t seg, int32_t slots, int32_t slots_per_lane, int32_t *jlo,
                                int32_t *rebase, uint32_t *row_off, int64_t *cells);
/* Stripe table of the wide-band kernel (k_dp_tile) for one segment (host logic, for inspection and tests): the lattice
 * columns 0..lX are cut into stripes of at most 64 * slots_per_lane columns, one wavefront sweeps a stripe anti-diagonal
 * by anti-diagonal.  stripes5[5 * k ..] = first column, columns, first / last anti-diagonal with band cells in the
 * stripe, index of its first row in the task's forward scratch (one row per anti-diagonal of a stripe); *rows = rows of
 * the segment.  Returns the number of stripes (stripes5 == NULL: only that), NPR_ERR_CAPACITY when cap is smaller. */
int32_t npr_plan_stripes(const npr_plan *pl, int32_t seg, int32_t slots_per_lane, int32_t *stripes5, int32_t cap, int64_t *rows);
/* MEA chain + cigar from sparse posteriors (stage a5.6).  Returns number of op pairs or NPR_ERR_*. */
int64_t npr_mea_cigar(int64_t lX, int64_t lY, const int32_t *x, const int32_t *y, const float *p, int64_t n,
                      double gap_gamma, double match_gamma, int32_t *ops, int64_t cap_pairs, double *score);
/* mean posterior over the M columns of a cigar (stage a5.7) */
int32_t npr_rescore(const int32_t *guide_ops, int64_t n_guide_ops, const int32_t *x, const int32_t *y,
                    const float *p, int64_t n, double *score);
/* Chaining of local hits (stage before the realigner, nanopore/analyses/utils.py:388-426 chainFn): hit k spans reference
 * [ref_start, ref_end] and signed read positions [read_start, read_end] (last aligned pair inclusive, reverse-strand
 * positions negated as AlignedPair.getSignedReadPos does), scores `score` (aligned pairs).  Writes the indices of the
 * highest-scoring co-linear chain, in chain order, to chain[0 .. return value); same chain as the reference's quadratic
 * scan, ties included, found with a sort and a windowed scan.  Host code. */
int64_t npr_chain_hits(int64_t n, const int64_t *ref_start, const int64_t *read_start, const int64_t *ref_end,
                       const int64_t *read_end, const uint8_t *reverse, const int64_t *score, int64_t max_gap, int64_t *chain);
/* The global alignment a chain stands for (mergeChainedAlignedReads, nanopore/analyses/utils.py:295-386): block k = a local
 * alignment whose first aligned pair is at reference position ref_pos[k] and position read_pos[k] of SEQ (its leading hard +
 * soft clips; the same on either strand), M / I / D operations ops[2 * ops_off[k] ..).  Unaligned reference / read bases
 * between, before and after the blocks become D / I operations, neighbours of one kind are merged; the result spans
 * ref_len x read_len (the asserts of utils.py:381-382).  Returns the number of (op, length) pairs written to out_ops,
 * NPR_ERR_INVALID when the blocks are out of order, overlap or run past the sequences (the reference's asserts),
 * NPR_ERR_CAPACITY when cap_pairs is too small (ops + 2 per block + 2 always suffices).  Host code. */
int64_t npr_chain_merge(int64_t n_blocks, const int64_t *ref_pos, const int64_t *read_pos, const int64_t *ops_off, const int32_t *ops,
                        int64_t ref_len, int64_t read_len, int32_t *out_ops, int64_t cap_pairs);
/* SAM CIGAR text of n op lists (CSR as returned by npr_batch_ops): what realignSamFile3TargetFn assigns to aR.cigar
 * and pysam prints (nanopore/analyses/utils.py:597-605), for a writer that splices 50 k records at once.  String i is
 * out[str_off[i] .. str_off[i+1]) (no terminator; "*" for an empty list).  out == NULL: only the offsets; returns the
 * total length, NPR_ERR_CAPACITY when cap is smaller, NPR_ERR_INVALID for an op outside M/I/D.  Threaded. */
int64_t npr_format_cigars(int64_t n, const int64_t *ops_off, const int32_t *ops, int64_t *str_off, char *out, int64_t cap);
/* The same from packed cigars, one 32-bit word per operation (length << 2 | op -- the form the ranks of a sharded job send
 * to rank 0, nanopore_amd/dist.py): list i = words[word_off[i] .. word_off[i] + n_ops[i]), the lists in any order and
 * anywhere in `words`, so the gathered payloads need no merge copy before the SAM is written. */
int64_t npr_format_cigars_packed(int64_t n, const int64_t *word_off, const int64_t *n_ops, const uint32_t *words, int64_t *str_off,
                                 char *out, int64_t cap);
/* The realigned SAM records themselves, as text: what realignSamFile3TargetFn's writer puts out for every record after
 * assigning the new cigar (nanopore/analyses/utils.py:591-609; pysam's AlignmentFile.write there), for a job that writes
 * 50 k records per rank at once.  Record i = QNAME \t FLAG \t RNAME \t POS \t MAPQ \t CIGAR \t * \t 0 \t 0 \t SEQ \t * \n with
 * QNAME = qnames[qname_off[i] .. qname_off[i+1]), RNAME = entry ref_index[i] of the rnames list, POS = pos[i] (1-based, as
 * printed), FLAG = flag[i] (NULL: 0), MAPQ = mapq[i] (NULL: 255), CIGAR = the packed list i as in npr_format_cigars_packed
 * ("*" when empty), SEQ = seq[seq_off[i] .. seq_off[i+1]) ("*" when empty).  Record i lands at out[rec_off[i] .. rec_off[i+1]).  out == NULL:
 * only the offsets; returns the total length, NPR_ERR_CAPACITY when cap is smaller, NPR_ERR_INVALID for a negative
 * number or an op outside M/I/D.  Threaded host code. */
int64_t npr_format_sam_records(int64_t n, const char *qnames, const int64_t *qname_off, const int32_t *flag, const char *rnames,
                               const int64_t *rname_off, const int32_t *ref_index, const int64_t *pos, const int32_t *mapq,
                               const int64_t *word_off, const int64_t *n_ops, const uint32_t *words, const char *seq, const int64_t *seq_off,
                               int64_t *rec_off, char *out, int64_t cap);
/* ASCII -> base codes 0..4 (A,C,G,T,N) */
void npr_encode_bases(const uint8_t *ascii, int64_t n, uint8_t *codes);

/* ---- bulk text ingest / splice: the file side of realignSamFile2TargetFn / realignSamFile3TargetFn ----
 * The reference walks the SAM file record by record in Python (pysam iterator + samIterator, nanopore/analyses/utils.py:287-293,
 * :563-570), builds one exonerate cigar per record (utils.py:168-180), and afterwards copies every record to the output with
 * only its CIGAR replaced (utils.py:591-609).  At 50 k records of 8 kb that loop is longer than the DP; these entry points do
 * the same over the whole text of the file (the caller maps or reads it), threaded, so that the records of a rank's shard go
 * from file bytes to the C ABI's batch arrays -- and the realigned cigars back into file bytes -- without a Python object
 * per record.  Host code; no GPU needed. */

/* Alignment lines of a SAM text: *header_end = offset of the first byte after the @-header lines; returns the number of
 * alignment lines (non-empty lines after the header).  span[2 * i], span[2 * i + 1] = [start, end) of line i without its
 * "\n" / "\r\n"; span == NULL: only count.  NPR_ERR_CAPACITY when cap (lines) is smaller. */
int64_t npr_sam_index(const char *text, int64_t len, int64_t *header_end, int64_t *span, int64_t cap);

/* Fields of n alignment lines (span as from npr_sam_index, possibly a sub-range: a rank's shard).  fields[i * NPR_SAM_COLS + c]:
 *   0 end of QNAME (it starts at span[2 * i])      1, 2  RNAME [start, end)       3, 4  CIGAR [start, end)
 *   5, 6 SEQ [start, end)                          7 FLAG     8 POS - 1 (0-based, pysam's aR.pos)     9 MAPQ
 *   10 tid = index of RNAME in the given name table (the @SQ order), -1 for "*" or a name not in it (column 15 says which)
 *   11, 12 [start, end) of the aligned part of SEQ in the text (soft clips cut off: aR.query, the sequence handed to
 *      cactus_realign, utils.py:570); empty for SEQ "*"
 *   13 operations M / I / D of the cigar (the guide: clips carry no operation, utils.py:173)
 *   14 reference bases the cigar consumes (aR.aend - aR.pos)
 *   15 NPR_OK; NPR_ERR_INVALID: fewer than 11 columns, a malformed number or cigar, or an operation outside M I D S H
 *      (the reference asserts `op in (0, 1, 2, 4, 5)`, utils.py:171; pysam's iterator raises on a line it cannot parse);
 *      NPR_SAM_NO_REFERENCE: RNAME is "*" -- the only records samIterator drops (utils.py:287-293);
 *      NPR_SAM_UNKNOWN_REFERENCE: RNAME names a sequence the table (the header's @SQ lines) does not have -- an error of
 *      the file (a SAM without its @SQ lines, a truncated header), never a record to drop silently
 * rnames / rname_off: the n_refs reference names, CSR.  Threaded. */
#define NPR_SAM_COLS 16
#define NPR_SAM_NO_REFERENCE 1
#define NPR_SAM_UNKNOWN_REFERENCE 2
int32_t npr_sam_parse(const char *text, const int64_t *span, int64_t n, const char *rnames, const int64_t *rname_off, int64_t n_refs,
                      int64_t *fields);
/* The guides of n parsed lines as the batch entry points take them: (op, length) pairs of the M / I / D operations of line i
 * at guide_ops[2 * guide_off[i] ..), guide_off = exclusive prefix sum of fields column 13 (made by the caller, n + 1 entries).
 * Lines whose column 15 is not NPR_OK are skipped. */
int32_t npr_sam_guides(const char *text, const int64_t *fields, int64_t n, const int64_t *guide_off, int32_t *guide_ops);
/* Output records: line i of the input with its CIGAR field replaced by the packed cigar i (one 32-bit word per operation,
 * length << 2 | op, list i = words[word_off[i] .. word_off[i] + n_ops[i]); "*" when empty), every other byte copied --
 * QNAME, FLAG, RNAME, POS, MAPQ, the mate fields, SEQ, QUAL and the tags exactly as the mapper wrote them, which is what
 * realignSamFile3TargetFn's `aR.cigar = ...; outputSam.write(aR)` produces (utils.py:597-605).  Each record ends in "\n".
 * Record i lands at out[rec_off[i] .. rec_off[i + 1]); out == NULL: only the offsets.  Returns the total length,
 * NPR_ERR_CAPACITY when cap is smaller, NPR_ERR_INVALID for an operation outside M / I / D.  Threaded. */
int64_t npr_sam_splice(const char *text, const int64_t *span, const int64_t *fields, int64_t n, const int64_t *word_off,
                       const int64_t *n_ops, const uint32_t *words, int64_t *rec_off, char *out, int64_t cap);

/* FASTA text (getFastaDictionary, utils.py:233-238): returns the number of records; rec[4 * k ..] = [start, end) of the
 * record's name (first word of the header line) and [start, end) of its sequence lines in the text, seq_len[k] = bases
 * (line ends and blanks not counted).  rec == NULL: only count.  NPR_ERR_CAPACITY when cap (records) is smaller. */
int64_t npr_fasta_index(const char *text, int64_t len, int64_t *rec, int64_t *seq_len, int64_t cap);
/* ... and the sequences themselves, contiguous: record k at out[seq_off[k] .. seq_off[k + 1]) (seq_off = prefix sum of
 * seq_len), line ends and blanks removed -- the reference table npr_batch_create takes.  Threaded over records. */
int32_t npr_fasta_pack(const char *text, const int64_t *rec, int64_t n, const int64_t *seq_off, uint8_t *out);
/* FASTQ text (getFastqDictionary, utils.py:240-245), four-line records: rec[4 * k ..] = [start, end) of the name (first
 * word after '@') and [start, end) of the sequence line.  rec == NULL: only count.  NPR_ERR_INVALID for a record that does
 * not start with '@' or whose third line does not start with '+'. */
int64_t npr_fastq_index(const char *text, int64_t len, int64_t *rec, int64_t cap);

/* npr_batch_create_at for reads that are NOT contiguous in memory: read i = read[read_begin[i] .. read_end[i]) -- e.g. the
 * aligned part of each record's SEQ inside the mapped SAM text (columns 11, 12 of npr_sam_parse), so the sequences go from
 * the file's bytes to the pinned staging buffer in one copy.  Everything else as npr_batch_create_at. */
int32_t npr_batch_create_spans(npr_ctx *ctx, const npr_params *params, int64_t n_reads, int64_t n_refs,
                               const uint8_t *ref, const int64_t *ref_off, const int32_t *ref_index,
                               const uint8_t *read, const int64_t *read_begin, const int64_t *read_end,
                               const int32_t *guide_ops, const int64_t *guide_off, const int64_t *guide_start,
                               const int32_t *model_slot, npr_batch **out);

#ifdef __cplusplus
}
#endif
#endif
