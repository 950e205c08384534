/*
 * realign_oracle.c -- CPU ORACLE (test infrastructure only; see realign_oracle.h header).
 *
 * Double-precision, log-space restatement of the cactus_realign / cPecan algorithm that the
 * reference shells out to at nanopore/analyses/utils.py:587 (realign),
 * nanopore/analyses/alignmentUncertainty.py:41 (rescore) and
 * nanopore/analyses/marginAlignSnpCaller.py:136-146 (all posteriors).
 * PARITY UNPINNED: the external program is absent from the snapshot; this file is
 * the normative definition used by this repository's parity tests.
 */
#include "realign_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NEG_INF (-INFINITY)
#define PROB_ONE 10000000LL /* posterior quantum used by the MEA stage (cPecan PAIR_ALIGNMENT_PROB_1) */

int32_t orc_version(void) { return 1; }

static inline int64_t i64min(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t i64max(int64_t a, int64_t b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------------
 * Band / segmentation.   SURVEY 8a rows a5.1 (anchors) and a5.2 (band, split).
 * ------------------------------------------------------------------------------------------ */

typedef struct {
    int64_t x, y;
} pt;

static int guide_ok(int64_t lX, int64_t lY, const int32_t *ops, int64_t nops) {
    int64_t sx = 0, sy = 0;
    for (int64_t i = 0; i < nops; i++) {
        int32_t op = ops[2 * i], len = ops[2 * i + 1];
        if (len < 0) return 0;
        if (op == ORC_OP_M) {
            sx += len;
            sy += len;
        } else if (op == ORC_OP_I) {
            sy += len;
        } else if (op == ORC_OP_D) {
            sx += len;
        } else {
            return 0;
        }
    }
    /* the realign input is a GLOBAL alignment: utils.py:381-382, :492-496 */
    return sx == lX && sy == lY;
}

static void seg_finish(orc_segment *s) {
    s->off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(s->D + 2));
    s->off[0] = 0;
    for (int64_t d = 0; d <= s->D; d++) s->off[d + 1] = s->off[d] + s->n[d];
    s->cells = s->off[s->D + 1];
}

/* band of one segment from its (segment-local) anchor points q[0..m], q[0]=(0,0), q[m]=(lX,lY) */
static void band_from_points(orc_segment *s, const pt *q, int64_t m, int64_t E) {
    int64_t lX = s->xe - s->xs, lY = s->ye - s->ys;
    s->D = lX + lY;
    s->lo = (int32_t *)malloc(sizeof(int32_t) * (size_t)(s->D + 1));
    s->n = (int32_t *)malloc(sizeof(int32_t) * (size_t)(s->D + 1));
    int64_t k = 0;
    for (int64_t d = 0; d <= s->D; d++) {
        /* interval k covers diagonals [d_k, d_{k+1}); the last interval also owns d_m */
        while (k + 1 < m && q[k + 1].x + q[k + 1].y <= d) k++;
        pt p = q[k], nx = q[k + 1 <= m ? k + 1 : m];
        /* rectangle [p.x,nx.x] x [p.y,nx.y] cut by the diagonal, expanded by E in xmy */
        int64_t low = i64max(2 * p.x - d, d - 2 * nx.y) - E;
        int64_t high = i64min(2 * nx.x - d, d - 2 * p.y) + E;
        low = i64max(low, i64max(-d, d - 2 * lY));
        high = i64min(high, i64min(d, 2 * lX - d));
        if ((low - d) & 1) low++;
        if ((high - d) & 1) high--;
        s->lo[d] = (int32_t)low;
        s->n[d] = (int32_t)((high - low) / 2 + 1);
    }
    seg_finish(s);
}

static orc_plan *plan_anchor(int64_t lX, int64_t lY, const int32_t *ops, int64_t nops, const orc_params *p) {
    int64_t trim = p->constraint_trim, E = p->diagonal_expansion, N = p->split_threshold;
    /* a5.1: every M column of the guide, minus `trim` columns at both ends of each gapless block,
     * becomes an anchor; as a lattice point the pair (x,y) of 0-based bases is (x+1,y+1). */
    int64_t cap = 2;
    for (int64_t i = 0; i < nops; i++)
        if (ops[2 * i] == ORC_OP_M) cap += ops[2 * i + 1];
    pt *P = (pt *)malloc(sizeof(pt) * (size_t)cap);
    int64_t np = 0;
    P[np++] = (pt){0, 0};
    int64_t x = 0, y = 0;
    for (int64_t i = 0; i < nops; i++) {
        int32_t op = ops[2 * i];
        int64_t len = ops[2 * i + 1];
        if (op == ORC_OP_M) {
            for (int64_t t = trim; t < len - trim; t++) P[np++] = (pt){x + t + 1, y + t + 1};
            x += len;
            y += len;
        } else if (op == ORC_OP_I) {
            y += len;
        } else {
            x += len;
        }
    }
    if (!(P[np - 1].x == lX && P[np - 1].y == lY)) P[np++] = (pt){lX, lY};

    /* a5.2: split where the unanchored rectangle between consecutive points exceeds N*N cells */
    orc_plan *pl = (orc_plan *)calloc(1, sizeof(orc_plan));
    int64_t segcap = 4;
    pl->seg = (orc_segment *)calloc((size_t)segcap, sizeof(orc_segment));
    pt *q = (pt *)malloc(sizeof(pt) * (size_t)(np + 2));
    int64_t m = 0; /* points collected for the current segment: q[0..m] */
    pt start = P[0];
    int32_t ragged_start = 0;
    q[0] = start;
    for (int64_t i = 0; i + 1 <= np - 1 + 0; i++) {
        pt a = P[i], b = P[i + 1];
        int64_t dX = b.x - a.x, dY = b.y - a.y;
        if (dX * dY > N * N) {
            int64_t hX = i64min(dX / 2, N), hY = i64min(dY / 2, N);
            pt A = {a.x + hX, a.y + hY}, B = {b.x - hX, b.y - hY};
            /* close the current segment at A (ragged end) */
            if (!(q[m].x == A.x && q[m].y == A.y)) q[++m] = A;
            if (pl->nseg == segcap) {
                segcap *= 2;
                pl->seg = (orc_segment *)realloc(pl->seg, sizeof(orc_segment) * (size_t)segcap);
            }
            orc_segment *s = &pl->seg[pl->nseg++];
            memset(s, 0, sizeof(*s));
            s->xs = start.x, s->ys = start.y, s->xe = A.x, s->ye = A.y;
            s->ragged_start = ragged_start, s->ragged_end = 1;
            for (int64_t k = 0; k <= m; k++) q[k].x -= start.x, q[k].y -= start.y;
            band_from_points(s, q, m, E);
            /* open the next one at B (ragged start) */
            start = B;
            ragged_start = 1;
            m = 0;
            q[0] = B;
            if (!(b.x == B.x && b.y == B.y)) q[++m] = b;
        } else {
            q[++m] = b;
        }
    }
    if (pl->nseg == segcap) {
        segcap += 1;
        pl->seg = (orc_segment *)realloc(pl->seg, sizeof(orc_segment) * (size_t)segcap);
    }
    orc_segment *s = &pl->seg[pl->nseg++];
    memset(s, 0, sizeof(*s));
    s->xs = start.x, s->ys = start.y, s->xe = lX, s->ye = lY;
    s->ragged_start = ragged_start, s->ragged_end = 0;
    for (int64_t k = 0; k <= m; k++) q[k].x -= start.x, q[k].y -= start.y;
    band_from_points(s, q, m, E);
    free(q);
    free(P);
    return pl;
}

/* fixed-width band (BASELINE configs "band=100 / 200"): xmy within +-W/2 of the guide path */
static orc_plan *plan_fixed(int64_t lX, int64_t lY, const int32_t *ops, int64_t nops, const orc_params *p) {
    int64_t h = p->fixed_width / 2;
    orc_plan *pl = (orc_plan *)calloc(1, sizeof(orc_plan));
    pl->nseg = 1;
    pl->seg = (orc_segment *)calloc(1, sizeof(orc_segment));
    orc_segment *s = &pl->seg[0];
    s->xs = 0, s->ys = 0, s->xe = lX, s->ye = lY;
    s->D = lX + lY;
    int64_t *c = (int64_t *)malloc(sizeof(int64_t) * (size_t)(s->D + 1));
    int64_t x = 0, y = 0;
    c[0] = 0;
    for (int64_t i = 0; i < nops; i++) {
        int32_t op = ops[2 * i];
        for (int64_t t = 0; t < ops[2 * i + 1]; t++) {
            if (op == ORC_OP_M) {
                c[x + y + 1] = x - y; /* the diagonal the match step jumps over */
                x++, y++;
            } else if (op == ORC_OP_D) {
                x++;
            } else {
                y++;
            }
            c[x + y] = x - y;
        }
    }
    s->lo = (int32_t *)malloc(sizeof(int32_t) * (size_t)(s->D + 1));
    s->n = (int32_t *)malloc(sizeof(int32_t) * (size_t)(s->D + 1));
    for (int64_t d = 0; d <= s->D; d++) {
        int64_t low = c[d] - h, high = c[d] + h;
        low = i64max(low, i64max(-d, d - 2 * lY));
        high = i64min(high, i64min(d, 2 * lX - d));
        if ((low - d) & 1) low++;
        if ((high - d) & 1) high--;
        s->lo[d] = (int32_t)low;
        s->n[d] = (int32_t)((high - low) / 2 + 1);
    }
    free(c);
    seg_finish(s);
    return pl;
}

orc_plan *orc_plan_build(int64_t lX, int64_t lY, const int32_t *ops, int64_t nops, const orc_params *p,
                         int32_t *status) {
    if (!guide_ok(lX, lY, ops, nops)) {
        if (status) *status = -1;
        return NULL;
    }
    if (status) *status = 0;
    if (p->band_mode == ORC_BAND_FIXED) return plan_fixed(lX, lY, ops, nops, p);
    return plan_anchor(lX, lY, ops, nops, p);
}

void orc_plan_free(orc_plan *pl) {
    if (!pl) return;
    for (int32_t i = 0; i < pl->nseg; i++) {
        free(pl->seg[i].lo);
        free(pl->seg[i].n);
        free(pl->seg[i].off);
    }
    free(pl->seg);
    free(pl);
}

int32_t orc_plan_nseg(const orc_plan *pl) { return pl->nseg; }

void orc_plan_seg_info(const orc_plan *pl, int32_t s, int64_t *info8) {
    const orc_segment *g = &pl->seg[s];
    info8[0] = g->xs, info8[1] = g->ys, info8[2] = g->xe, info8[3] = g->ye;
    info8[4] = g->ragged_start, info8[5] = g->ragged_end, info8[6] = g->D, info8[7] = g->cells;
}

void orc_plan_seg_band(const orc_plan *pl, int32_t s, int32_t *lo, int32_t *n) {
    const orc_segment *g = &pl->seg[s];
    memcpy(lo, g->lo, sizeof(int32_t) * (size_t)(g->D + 1));
    memcpy(n, g->n, sizeof(int32_t) * (size_t)(g->D + 1));
}

/* ------------------------------------------------------------------------------------------
 * Model tables.  HMM layout: SURVEY 8a row a7.
 * ------------------------------------------------------------------------------------------ */

typedef struct {
    double lT[5][5];  /* log transition from -> to */
    double lEm[5][5]; /* log match emission [x][y], index 4 = N */
    double lEx[5][5]; /* log gap-X emission [state][x]  (states 1,3 used) */
    double lEy[5][5]; /* log gap-Y emission [state][y]  (states 2,4 used) */
    double lStart[2][5], lEnd[2][5];
} model64;

static double slog(double v) { return v > 0.0 ? log(v) : NEG_INF; }

static void model64_init(model64 *m, const orc_hmm *h) {
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++) m->lT[a][b] = slog(h->T[a * 5 + b]);
    for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++) m->lEm[x][y] = (x < 4 && y < 4) ? slog(h->E[x * 4 + y]) : log(1.0 / 16.0);
    for (int s = 0; s < 5; s++) {
        for (int x = 0; x < 5; x++) {
            double ex = 0.0, ey = 0.0;
            if (x < 4) {
                for (int y = 0; y < 4; y++) ex += h->E[s * 16 + x * 4 + y]; /* marginal over the read base */
                for (int y = 0; y < 4; y++) ey += h->E[s * 16 + y * 4 + x]; /* marginal over the ref base */
            } else {
                ex = ey = 0.25;
            }
            m->lEx[s][x] = slog(ex);
            m->lEy[s][x] = slog(ey);
        }
    }
    /* global ends: start in match; end weighted by the transition back to match.
     * ragged ends (split points): start inside a long gap; end by opening / extending one. */
    for (int s = 0; s < 5; s++) {
        m->lStart[0][s] = s == 0 ? 0.0 : NEG_INF;
        m->lStart[1][s] = (s == 3 || s == 4) ? 0.0 : NEG_INF;
        m->lEnd[0][s] = m->lT[s][0];
    }
    m->lEnd[1][0] = m->lT[0][3];
    m->lEnd[1][1] = m->lT[0][3];
    m->lEnd[1][2] = m->lT[0][4];
    m->lEnd[1][3] = m->lT[3][3];
    m->lEnd[1][4] = m->lT[4][4];
}

/* log(e^a + e^b).  Two kinds (orc_set_logadd_kind):
 *   0  exact: max + log1p(exp(-|a - b|)) -- the normative arithmetic of this oracle;
 *   1  cPecan's: max + lookup(|a - b|) with the probcons-style four-piece cubic fit of log(1 + e^-t)... written there as
 *      min + lookup(max - min), lookup(t) ~ log(1 + e^t) on [0, 7.5], and just max beyond 7.5.  [RECALLED, SURVEY.md
 *      Appendix A: cactus / cPecan are absent from the snapshot, so the coefficients below are from memory of the public
 *      probcons source (ScoreType.h LOOKUP) that cPecan's pairwiseAligner.c copies; tests/test_oracle.py checks that they
 *      do approximate log(1 + e^t) to ~1e-4, which is all that is claimed.]  Used only to MEASURE how far the reference's
 *      own approximation could move a posterior or a cigar; never the checker of a parity test. */
static int g_logadd_kind = 0;
void orc_set_logadd_kind(int32_t kind) { g_logadd_kind = kind; }
int32_t orc_get_logadd_kind(void) { return g_logadd_kind; }

static inline double logadd_lookup(double t) { /* t = max - min in [0, 7.5] */
    if (t <= 1.00) return ((-0.009350833524763 * t + 0.130659527668286) * t + 0.498799810682272) * t + 0.693203116424741;
    if (t <= 2.50) return ((-0.014532321752540 * t + 0.139942324101744) * t + 0.495635523139337) * t + 0.692140569840976;
    if (t <= 4.50) return ((-0.004605031767994 * t + 0.063427417320019) * t + 0.695956496475118) * t + 0.514272634594009;
    return ((-0.000458661602210 * t + 0.009695946122598) * t + 0.930734667215156) * t + 0.168037164329057;
}

double orc_logadd(double a, double b); /* exported for the tests */

static inline double logadd(double a, double b) {
    if (a == NEG_INF) return b;
    if (b == NEG_INF) return a;
    if (g_logadd_kind == 1) {
        const double lo = a < b ? a : b, hi = a < b ? b : a;
        return hi - lo >= 7.5 ? hi : lo + logadd_lookup(hi - lo);
    }
    return a > b ? a + log1p(exp(b - a)) : b + log1p(exp(a - b));
}
double orc_logadd(double a, double b) { return logadd(a, b); }

/* Per-thread scratch that only grows: the forward array of a 10 kb x band-200 read is 160 MB, and a fresh
 * malloc/free of that per read turns the multi-threaded baseline into a page-fault benchmark. */
static _Thread_local double *tl_buf[2] = {NULL, NULL};
static _Thread_local size_t tl_cap[2] = {0, 0};
static double *scratch(int which, size_t n) {
    if (tl_cap[which] < n) {
        free(tl_buf[which]);
        tl_buf[which] = (double *)malloc(sizeof(double) * n);
        tl_cap[which] = tl_buf[which] ? n : 0;
    }
    return tl_buf[which];
}

/* move type of the destination state: 0 = diagonal (match), 1 = x only, 2 = y only */
static const int MOVE[5] = {0, 1, 2, 1, 2};

static inline int64_t cell_index(const int32_t *lo, const int32_t *n, const int64_t *off, int64_t D, int64_t d,
                                 int64_t xmy) {
    if (d < 0 || d > D) return -1;
    int64_t j = xmy - lo[d];
    if (j < 0 || (j & 1)) return -1;
    j >>= 1;
    if (j >= n[d]) return -1;
    return off[d] + j;
}

int32_t orc_fb_f64(const orc_hmm *h, const uint8_t *X, int64_t lX, const uint8_t *Y, int64_t lY,
                   const int32_t *lo, const int32_t *n, int32_t ragged_start, int32_t ragged_end,
                   double threshold, double *total_ll, double *total_ll_bwd, double *Fm, double *Bm,
                   double *Fall, double *Ball, int32_t *px, int32_t *py, double *pp, int64_t cap,
                   int64_t *npairs) {
    model64 m;
    model64_init(&m, h);
    const int64_t D = lX + lY;
    int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(D + 2));
    off[0] = 0;
    for (int64_t d = 0; d <= D; d++) off[d + 1] = off[d] + n[d];
    const int64_t cells = off[D + 1];
    double *F = scratch(0, 5 * (size_t)cells);
    int32_t rc = 0;

    /* ---- forward (a5.3) ---- */
    for (int64_t d = 0; d <= D; d++) {
        for (int64_t j = 0; j < n[d]; j++) {
            const int64_t xmy = lo[d] + 2 * j;
            const int64_t x = (d + xmy) / 2, y = (d - xmy) / 2;
            double *c = F + 5 * (off[d] + j);
            if (x < 0 || y < 0 || x > lX || y > lY) {
                for (int s = 0; s < 5; s++) c[s] = NEG_INF;
                continue;
            }
            if (d == 0) {
                for (int s = 0; s < 5; s++) c[s] = m.lStart[ragged_start ? 1 : 0][s];
                continue;
            }
            const int64_t iM = (x > 0 && y > 0) ? cell_index(lo, n, off, D, d - 2, xmy) : -1;
            const int64_t iL = (x > 0) ? cell_index(lo, n, off, D, d - 1, xmy - 1) : -1; /* (x-1,y) */
            const int64_t iU = (y > 0) ? cell_index(lo, n, off, D, d - 1, xmy + 1) : -1; /* (x,y-1) */
            for (int t = 0; t < 5; t++) {
                const int64_t ip = MOVE[t] == 0 ? iM : (MOVE[t] == 1 ? iL : iU);
                if (ip < 0) {
                    c[t] = NEG_INF;
                    continue;
                }
                const double *pc = F + 5 * ip;
                double acc = NEG_INF;
                for (int s = 0; s < 5; s++) acc = logadd(acc, pc[s] + m.lT[s][t]);
                double e = MOVE[t] == 0 ? m.lEm[X[x - 1]][Y[y - 1]]
                                        : (MOVE[t] == 1 ? m.lEx[t][X[x - 1]] : m.lEy[t][Y[y - 1]]);
                c[t] = acc + e;
            }
        }
    }
    double tot = NEG_INF;
    {
        const int64_t ie = cell_index(lo, n, off, D, D, lX - lY);
        if (ie >= 0)
            for (int s = 0; s < 5; s++) tot = logadd(tot, F[5 * ie + s] + m.lEnd[ragged_end ? 1 : 0][s]);
    }
    if (total_ll) *total_ll = tot;
    if (Fm)
        for (int64_t i = 0; i < cells; i++) Fm[i] = F[5 * i];
    if (Fall) memcpy(Fall, F, sizeof(double) * 5 * (size_t)cells);
    if (tot == NEG_INF) rc = -2;

    /* ---- backward (a5.4) ---- */
    int64_t np = 0;
    if (rc == 0) {
        int64_t wmax = 0;
        for (int64_t d = 0; d <= D; d++)
            if (n[d] > wmax) wmax = n[d];
        double *ring = (double *)malloc(sizeof(double) * 5 * 3 * (size_t)wmax);
        double *Bmatch = scratch(1, (size_t)cells);
        double totb = NEG_INF;
        for (int64_t d = D; d >= 0; d--) {
            double *cur = ring + 5 * wmax * (d % 3);
            for (int64_t j = 0; j < n[d]; j++) {
                const int64_t xmy = lo[d] + 2 * j;
                const int64_t x = (d + xmy) / 2, y = (d - xmy) / 2;
                double *c = cur + 5 * j;
                if (x < 0 || y < 0 || x > lX || y > lY) {
                    for (int s = 0; s < 5; s++) c[s] = NEG_INF;
                } else if (d == D) {
                    for (int s = 0; s < 5; s++) c[s] = m.lEnd[ragged_end ? 1 : 0][s];
                } else {
                    /* successors: (x+1,y+1) on d+2, (x+1,y) and (x,y+1) on d+1 */
                    const int64_t jM = (x < lX && y < lY) ? cell_index(lo, n, off, D, d + 2, xmy) : -1;
                    const int64_t jX = (x < lX) ? cell_index(lo, n, off, D, d + 1, xmy + 1) : -1;
                    const int64_t jY = (y < lY) ? cell_index(lo, n, off, D, d + 1, xmy - 1) : -1;
                    const double *bM = jM >= 0 ? ring + 5 * wmax * ((d + 2) % 3) + 5 * (jM - off[d + 2]) : NULL;
                    const double *bX = jX >= 0 ? ring + 5 * wmax * ((d + 1) % 3) + 5 * (jX - off[d + 1]) : NULL;
                    const double *bY = jY >= 0 ? ring + 5 * wmax * ((d + 1) % 3) + 5 * (jY - off[d + 1]) : NULL;
                    for (int s = 0; s < 5; s++) {
                        double acc = NEG_INF;
                        for (int t = 0; t < 5; t++) {
                            const double *bp = MOVE[t] == 0 ? bM : (MOVE[t] == 1 ? bX : bY);
                            if (!bp) continue;
                            double e = MOVE[t] == 0 ? m.lEm[X[x]][Y[y]]
                                                    : (MOVE[t] == 1 ? m.lEx[t][X[x]] : m.lEy[t][Y[y]]);
                            acc = logadd(acc, m.lT[s][t] + e + bp[t]);
                        }
                        c[s] = acc;
                    }
                }
                const int64_t ic = off[d] + j;
                Bmatch[ic] = c[0];
                if (Ball)
                    for (int s = 0; s < 5; s++) Ball[5 * ic + s] = c[s];
                if (d == 0 && x == 0 && y == 0)
                    for (int s = 0; s < 5; s++) totb = logadd(totb, m.lStart[ragged_start ? 1 : 0][s] + c[s]);
            }
        }
        if (total_ll_bwd) *total_ll_bwd = totb;
        if (Bm) memcpy(Bm, Bmatch, sizeof(double) * (size_t)cells);
        free(ring);
        /* ---- posterior match probabilities (a5.5), diagonal ascending, xmy ascending ---- */
        for (int64_t d = 2; d <= D && (px || npairs); d++) {
            for (int64_t j = 0; j < n[d]; j++) {
                const int64_t xmy = lo[d] + 2 * j;
                const int64_t x = (d + xmy) / 2, y = (d - xmy) / 2;
                if (x < 1 || y < 1 || x > lX || y > lY) continue;
                const int64_t ic = off[d] + j;
                const double lp = F[5 * ic] + Bmatch[ic] - tot;
                if (lp == NEG_INF || lp != lp) continue;
                const double pr = exp(lp);
                if (pr >= threshold) {
                    if (px) {
                        if (np < cap) {
                            px[np] = (int32_t)(x - 1);
                            py[np] = (int32_t)(y - 1);
                            pp[np] = pr;
                        } else {
                            rc = -3;
                        }
                    }
                    np++;
                }
            }
        }
    }
    if (npairs) *npairs = np;
    free(off);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Baum-Welch expectations.
 * ------------------------------------------------------------------------------------------ */
int32_t orc_expectations_f64(const orc_hmm *h, const uint8_t *X, int64_t lX, const uint8_t *Y, int64_t lY,
                             const int32_t *lo, const int32_t *n, int32_t ragged_start, int32_t ragged_end,
                             double *T_exp, double *E_exp, double *total_ll) {
    const int64_t D = lX + lY;
    int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(D + 2));
    off[0] = 0;
    for (int64_t d = 0; d <= D; d++) off[d + 1] = off[d] + n[d];
    const int64_t cells = off[D + 1];
    double *F = (double *)malloc(sizeof(double) * 5 * (size_t)cells);
    double *B = (double *)malloc(sizeof(double) * 5 * (size_t)cells);
    double tot = 0.0, totb = 0.0;
    int32_t rc = orc_fb_f64(h, X, lX, Y, lY, lo, n, ragged_start, ragged_end, 2.0, &tot, &totb, NULL, NULL, F, B, NULL,
                            NULL, NULL, 0, NULL);
    if (total_ll) *total_ll = tot;
    if (rc == 0) {
        model64 m;
        model64_init(&m, h);
        for (int64_t d = 1; d <= D; d++)
            for (int64_t j = 0; j < n[d]; j++) {
                const int64_t xmy = lo[d] + 2 * j, x = (d + xmy) / 2, y = (d - xmy) / 2;
                if (x < 0 || y < 0 || x > lX || y > lY) continue;
                const double *bc = B + 5 * (off[d] + j);
                const int64_t iM = (x > 0 && y > 0) ? cell_index(lo, n, off, D, d - 2, xmy) : -1;
                const int64_t iL = (x > 0) ? cell_index(lo, n, off, D, d - 1, xmy - 1) : -1;
                const int64_t iU = (y > 0) ? cell_index(lo, n, off, D, d - 1, xmy + 1) : -1;
                for (int t = 0; t < 5; t++) {
                    const int64_t ip = MOVE[t] == 0 ? iM : (MOVE[t] == 1 ? iL : iU);
                    if (ip < 0 || bc[t] == NEG_INF) continue;
                    const int cx = x > 0 ? X[x - 1] : 4, cy = y > 0 ? Y[y - 1] : 4;
                    const double e = MOVE[t] == 0 ? m.lEm[cx][cy] : (MOVE[t] == 1 ? m.lEx[t][cx] : m.lEy[t][cy]);
                    double into = 0.0;
                    for (int s = 0; s < 5; s++) {
                        const double lp = F[5 * ip + s] + m.lT[s][t] + e + bc[t] - tot;
                        if (lp == NEG_INF || lp != lp) continue;
                        const double p = exp(lp);
                        T_exp[s * 5 + t] += p;
                        into += p;
                    }
                    if (MOVE[t] == 0) {
                        if (cx < 4 && cy < 4) E_exp[cx * 4 + cy] += into;
                    } else if (MOVE[t] == 1) {
                        if (cx < 4)
                            for (int q = 0; q < 4; q++) E_exp[t * 16 + cx * 4 + q] += 0.25 * into;
                    } else {
                        if (cy < 4)
                            for (int q = 0; q < 4; q++) E_exp[t * 16 + q * 4 + cy] += 0.25 * into;
                    }
                }
            }
    }
    free(F);
    free(B);
    free(off);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * MEA chain + cigar.  SURVEY 8a row a5.6.
 * ------------------------------------------------------------------------------------------ */

typedef struct {
    int32_t x, y;
    int64_t P, w;
} wpair;

static int cmp_wpair(const void *a, const void *b) {
    const wpair *p = (const wpair *)a, *q = (const wpair *)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    return 0;
}

typedef struct {
    int64_t S;
    int64_t idx;
} fen;

static inline int fen_better(fen a, fen b) { /* a strictly better than b */
    return a.S > b.S || (a.S == b.S && a.idx > b.idx);
}

static int64_t emit_op(int32_t *out, int64_t nout, int64_t cap, int32_t op, int64_t len) {
    if (len <= 0) return nout;
    if (nout > 0 && out[2 * (nout - 1)] == op) {
        out[2 * (nout - 1) + 1] += (int32_t)len;
        return nout;
    }
    if (nout >= cap) return -3;
    out[2 * nout] = op;
    out[2 * nout + 1] = (int32_t)len;
    return nout + 1;
}

int64_t orc_mea_cigar(int64_t lX, int64_t lY, const int32_t *px, const int32_t *py, const double *pp,
                      int64_t npairs, double gap_gamma, double match_gamma, int32_t *out_ops,
                      int64_t cap_ops, double *score, int32_t brute_force) {
    /* quantise posteriors; per-position gap mass = 1 - sum of match posteriors in that row / column */
    int64_t *gx = (int64_t *)calloc((size_t)(lX + 1), sizeof(int64_t));
    int64_t *gy = (int64_t *)calloc((size_t)(lY + 1), sizeof(int64_t));
    wpair *W = (wpair *)malloc(sizeof(wpair) * (size_t)(npairs + 1));
    for (int64_t i = 0; i < npairs; i++) {
        int64_t P = (int64_t)floor(pp[i] * (double)PROB_ONE);
        W[i].x = px[i], W[i].y = py[i], W[i].P = P;
        gx[px[i]] += P;
        gy[py[i]] += P;
    }
    for (int64_t i = 0; i < lX; i++) gx[i] = i64max(0, PROB_ONE - gx[i]);
    for (int64_t i = 0; i < lY; i++) gy[i] = i64max(0, PROB_ONE - gy[i]);
    const int64_t wmin = (int64_t)floor(match_gamma * (double)PROB_ONE);
    int64_t k = 0;
    for (int64_t i = 0; i < npairs; i++) {
        int64_t w = W[i].P - (int64_t)floor(gap_gamma * (double)(gx[W[i].x] + gy[W[i].y]));
        if (w > wmin) {
            W[k] = W[i];
            W[k].w = w;
            k++;
        }
    }
    qsort(W, (size_t)k, sizeof(wpair), cmp_wpair);

    int64_t *S = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k + 1));
    int64_t *pred = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k + 1));
    if (brute_force) {
        for (int64_t i = 0; i < k; i++) {
            fen best = {0, -1};
            for (int64_t j = 0; j < i; j++)
                if (W[j].x < W[i].x && W[j].y < W[i].y) {
                    fen c = {S[j], j};
                    if (fen_better(c, best)) best = c;
                }
            S[i] = W[i].w + best.S;
            pred[i] = best.idx;
        }
    } else {
        /* Fenwick tree over y holding the best (score, index) among inserted pairs with y' <= key */
        fen *tr = (fen *)malloc(sizeof(fen) * (size_t)(lY + 2));
        for (int64_t i = 0; i <= lY + 1; i++) tr[i] = (fen){0, -1};
        int64_t g0 = 0;
        while (g0 < k) {
            int64_t g1 = g0;
            while (g1 < k && W[g1].x == W[g0].x) g1++;
            for (int64_t i = g0; i < g1; i++) { /* query y' < y  <=> tree positions 1..y */
                fen best = {0, -1};
                for (int64_t t = W[i].y; t > 0; t -= t & (-t))
                    if (fen_better(tr[t], best)) best = tr[t];
                S[i] = W[i].w + best.S;
                pred[i] = best.idx;
            }
            for (int64_t i = g0; i < g1; i++) { /* insert at position y+1 */
                fen c = {S[i], i};
                for (int64_t t = W[i].y + 1; t <= lY + 1; t += t & (-t))
                    if (fen_better(c, tr[t])) tr[t] = c;
            }
            g0 = g1;
        }
        free(tr);
    }
    fen best = {0, -1};
    for (int64_t i = 0; i < k; i++) {
        fen c = {S[i], i};
        if (fen_better(c, best)) best = c;
    }
    /* trace back */
    int64_t nchain = 0;
    for (int64_t i = best.idx; i >= 0; i = pred[i]) nchain++;
    int64_t *chain = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nchain + 1));
    {
        int64_t t = nchain;
        for (int64_t i = best.idx; i >= 0; i = pred[i]) chain[--t] = i;
    }
    /* ops: unaligned reference bases first (D), then unaligned read bases (I), then the match */
    int64_t nout = 0, lastx = -1, lasty = -1, sumP = 0;
    for (int64_t c = 0; c < nchain && nout >= 0; c++) {
        const wpair *w = &W[chain[c]];
        nout = emit_op(out_ops, nout, cap_ops, ORC_OP_D, w->x - lastx - 1);
        if (nout >= 0) nout = emit_op(out_ops, nout, cap_ops, ORC_OP_I, w->y - lasty - 1);
        if (nout >= 0) nout = emit_op(out_ops, nout, cap_ops, ORC_OP_M, 1);
        lastx = w->x, lasty = w->y;
        sumP += w->P;
    }
    if (nout >= 0) nout = emit_op(out_ops, nout, cap_ops, ORC_OP_D, lX - 1 - lastx);
    if (nout >= 0) nout = emit_op(out_ops, nout, cap_ops, ORC_OP_I, lY - 1 - lasty);
    if (score) *score = nchain > 0 ? (double)sumP / ((double)nchain * (double)PROB_ONE) : 0.0;
    free(chain);
    free(S);
    free(pred);
    free(W);
    free(gx);
    free(gy);
    return nout;
}

/* a5.7: mean posterior over the guide's M columns; columns below the threshold count as 0 */
typedef struct {
    int32_t x, y;
    double p;
} ppair;

static int cmp_ppair(const void *a, const void *b) {
    const ppair *p = (const ppair *)a, *q = (const ppair *)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    if (p->y != q->y) return p->y < q->y ? -1 : 1;
    return 0;
}

double orc_rescore(const int32_t *ops, int64_t nops, const int32_t *px, const int32_t *py, const double *pp,
                   int64_t npairs) {
    ppair *Q = (ppair *)malloc(sizeof(ppair) * (size_t)(npairs + 1));
    for (int64_t i = 0; i < npairs; i++) Q[i] = (ppair){px[i], py[i], pp[i]};
    qsort(Q, (size_t)npairs, sizeof(ppair), cmp_ppair);
    int64_t x = 0, y = 0, nm = 0;
    double sum = 0.0;
    for (int64_t i = 0; i < nops; i++) {
        int32_t op = ops[2 * i];
        int64_t len = ops[2 * i + 1];
        if (op == ORC_OP_M) {
            for (int64_t t = 0; t < len; t++) {
                ppair key = {(int32_t)(x + t), (int32_t)(y + t), 0.0};
                ppair *f = (ppair *)bsearch(&key, Q, (size_t)npairs, sizeof(ppair), cmp_ppair);
                if (f) sum += f->p;
            }
            nm += len;
            x += len, y += len;
        } else if (op == ORC_OP_I) {
            y += len;
        } else {
            x += len;
        }
    }
    free(Q);
    return nm > 0 ? sum / (double)nm : 0.0;
}

/* ------------------------------------------------------------------------------------------
 * Whole read.
 * ------------------------------------------------------------------------------------------ */

int32_t orc_realign_read(const orc_hmm *h, const orc_params *p, int32_t precision, const uint8_t *X,
                         int64_t lX, const uint8_t *Y, int64_t lY, const int32_t *guide_ops,
                         int64_t n_guide_ops, int32_t *out_ops, int64_t cap_ops, int32_t *px, int32_t *py,
                         double *pp, int64_t cap_pairs, orc_read_result *res) {
    return orc_realign_read_arith(h, p, precision, NULL, 0, X, lX, Y, lY, guide_ops, n_guide_ops, out_ops, cap_ops, px, py,
                                  pp, cap_pairs, res);
}

int32_t orc_realign_read_arith(const orc_hmm *h, const orc_params *p, int32_t precision, const int32_t *seg_arith,
                               int64_t n_seg_arith, const uint8_t *X, int64_t lX, const uint8_t *Y, int64_t lY,
                               const int32_t *guide_ops, int64_t n_guide_ops, int32_t *out_ops, int64_t cap_ops,
                               int32_t *px, int32_t *py, double *pp, int64_t cap_pairs, orc_read_result *res) {
    memset(res, 0, sizeof(*res));
    int32_t st = 0;
    orc_plan *pl = orc_plan_build(lX, lY, guide_ops, n_guide_ops, p, &st);
    if (!pl) {
        res->status = st;
        return st;
    }
    int own = 0;
    if (!px) {
        own = 1;
        cap_pairs = 4 * (lX < lY ? lX : lY) + 1024;
        px = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap_pairs);
        py = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap_pairs);
        pp = (double *)malloc(sizeof(double) * (size_t)cap_pairs);
    }
    int64_t np = 0;
    for (int32_t s = 0; s < pl->nseg && st == 0; s++) {
        const orc_segment *g = &pl->seg[s];
        int64_t got = 0;
        double ll = 0.0;
        if (precision == 0) {
            st = orc_fb_f64(h, X + g->xs, g->xe - g->xs, Y + g->ys, g->ye - g->ys, g->lo, g->n, g->ragged_start,
                            g->ragged_end, p->posterior_threshold, &ll, NULL, NULL, NULL, NULL, NULL, px + np,
                            py + np, pp + np, cap_pairs - np, &got);
        } else {
            float tm = 0.f, bm = 0.f;
            int32_t te = 0, be = 0;
            float *pf = (float *)malloc(sizeof(float) * (size_t)(cap_pairs - np + 1));
            const int rs = seg_arith && s < n_seg_arith && seg_arith[s] == 1;
            st = (rs ? orc_fb_f32_rs : orc_fb_f32)(h, X + g->xs, g->xe - g->xs, Y + g->ys, g->ye - g->ys, g->lo, g->n,
                                                   g->ragged_start, g->ragged_end, (float)p->posterior_threshold, &tm, &te,
                                                   &bm, &be, NULL, NULL, NULL, NULL, px + np, py + np, pf, cap_pairs - np,
                                                   &got);
            if (st == 1) st = 0; /* orc_fb_f32_rs: the range certificate failed (the caller asked for this arithmetic anyway) */
            if (st == 0)
                for (int64_t i = 0; i < got; i++) pp[np + i] = (double)pf[i];
            free(pf);
            ll = (log2((double)tm) + (double)te) * M_LN2;
        }
        if (st == 0) {
            for (int64_t i = np; i < np + got; i++) {
                px[i] += (int32_t)g->xs;
                py[i] += (int32_t)g->ys;
            }
            np += got;
            res->total_ll += ll;
            res->cells += g->cells;
        }
    }
    orc_plan_free(pl);
    res->npairs = np;
    if (st == 0) {
        if (p->mode == ORC_MODE_RESCORE_ORIGINAL) {
            res->score = orc_rescore(guide_ops, n_guide_ops, px, py, pp, np);
            int64_t k = 0;
            for (int64_t i = 0; i < n_guide_ops && out_ops; i++) {
                if (guide_ops[2 * i + 1] <= 0) continue;
                if (k >= cap_ops) {
                    st = -3;
                    break;
                }
                out_ops[2 * k] = guide_ops[2 * i];
                out_ops[2 * k + 1] = guide_ops[2 * i + 1];
                k++;
            }
            res->nops = k;
        } else {
            int64_t no = orc_mea_cigar(lX, lY, px, py, pp, np, p->gap_gamma, p->match_gamma, out_ops, cap_ops,
                                       &res->score, 0);
            if (no < 0)
                st = (int32_t)no;
            else
                res->nops = no;
        }
    }
    if (own) {
        free(px);
        free(py);
        free(pp);
    }
    res->status = st;
    return st;
}

int32_t orc_realign_batch(const orc_hmm *h, const orc_params *p, int32_t precision, int64_t nreads,
                          const uint8_t *X, const int64_t *x_off, const uint8_t *Y, const int64_t *y_off,
                          const int32_t *guide_ops, const int64_t *g_off, int32_t *out_ops, const int64_t *o_off,
                          int64_t *out_nops, double *out_score, double *out_ll, int64_t *out_cells,
                          int32_t *out_status, int32_t threads) {
    return orc_realign_batch_arith(h, p, precision, NULL, NULL, nreads, X, x_off, Y, y_off, guide_ops, g_off, out_ops, o_off,
                                   out_nops, out_score, out_ll, out_cells, out_status, threads);
}

int32_t orc_realign_batch_arith(const orc_hmm *h, const orc_params *p, int32_t precision, const int64_t *seg_off,
                                const int32_t *seg_arith, int64_t nreads, const uint8_t *X, const int64_t *x_off,
                                const uint8_t *Y, const int64_t *y_off, const int32_t *guide_ops, const int64_t *g_off,
                                int32_t *out_ops, const int64_t *o_off, int64_t *out_nops, double *out_score,
                                double *out_ll, int64_t *out_cells, int32_t *out_status, int32_t threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int64_t i = 0; i < nreads; i++) {
        orc_read_result r;
        orc_realign_read_arith(h, p, precision, seg_arith && seg_off ? seg_arith + seg_off[i] : NULL,
                               seg_arith && seg_off ? seg_off[i + 1] - seg_off[i] : 0, X + x_off[i], x_off[i + 1] - x_off[i],
                               Y + y_off[i], y_off[i + 1] - y_off[i], guide_ops + 2 * g_off[i], g_off[i + 1] - g_off[i],
                               out_ops + 2 * o_off[i], o_off[i + 1] - o_off[i], NULL, NULL, NULL, 0, &r);
        out_nops[i] = r.nops;
        out_score[i] = r.score;
        out_ll[i] = r.total_ll;
        out_cells[i] = r.cells;
        out_status[i] = r.status;
    }
    return 0;
}
