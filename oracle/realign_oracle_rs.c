/*
 * realign_oracle_rs.c -- CPU ORACLE, fp32 "mirror" of the device's ROW-SCALED arithmetic.
 * TEST INFRASTRUCTURE ONLY (see realign_oracle.h).
 *
 * The one-wavefront frame kernels (k_dp_rs, nanopore_amd/csrc/npr_rs.h) evaluate the same forward / backward
 * recurrences as orc_fb_f64 (call sites nanopore/analyses/utils.py:587,
 * alignmentUncertainty.py:41, marginAlignSnpCaller.py:136-146) as the classic scaled HMM recurrence: the cells of an
 * anti-diagonal are plain fp32 values relative to one binary exponent shared by the whole row.  The two rows the
 * recurrence reads always carry the same exponent; after every RS_K-th anti-diagonal (d % RS_K == 0, RS_K = 16) both are
 * multiplied by the power of two that brings their largest value into [2^84, 2^85) -- near the top of fp32's range, so that
 * a cell stays a normal number down to 211 binary orders below its row's maximum.  Cells outside the band are exact
 * zeros.  The forward sweep keeps, per row, the match values as they were when the row was finished together with
 * the exponent they were relative to; the backward sweep forms F * (B * 2^(eF + eB - eTot)) / totMant.
 *
 * This file restates that sequence operation by operation (every operation is an IEEE-754 single-rounding mul / fma /
 * ldexp / frexp, denormals included); the parity tests require the GPU results to be IDENTICAL to it and require it
 * to agree with the double-precision log-space oracle within the stated tolerance.
 *
 * Compile with -ffp-contract=off: every fused multiply-add below is an explicit fmaf().
 */
#include "realign_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define RS_K 16 /* nanopore_amd/csrc/npr_device.h: NPR_RS_K */
#define RS_TOP 85 /* NPR_RS_TOP */
#define RS_S_LIMIT (126 - 60 - (RS_TOP + 6) - 1) /* NPR_RS_S_LIMIT: see nanopore_amd/csrc/npr_device.h */
#define E_DEAD (-(1 << 28))

typedef struct {
    float v[5]; /* 0 match, 1 shortGapX, 2 shortGapY, 3 longGapX, 4 longGapY */
} rcell;

typedef struct {
    float T[5][5];
    float em[5][5];
    float ex[5][5]; /* [state][x] */
    float ey[5][5]; /* [state][y] */
    float start[2][5], end[2][5];
} rmodel;

static void rmodel_init(rmodel *m, const orc_hmm *h) {
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++) m->T[a][b] = (float)h->T[a * 5 + b];
    for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++) m->em[x][y] = (x < 4 && y < 4) ? (float)h->E[x * 4 + y] : 0.0625f;
    for (int s = 0; s < 5; s++)
        for (int x = 0; x < 5; x++) {
            double ex = 0.0, ey = 0.0;
            if (x < 4) {
                for (int y = 0; y < 4; y++) ex += h->E[s * 16 + x * 4 + y];
                for (int y = 0; y < 4; y++) ey += h->E[s * 16 + y * 4 + x];
            } else {
                ex = ey = 0.25;
            }
            m->ex[s][x] = (float)ex;
            m->ey[s][x] = (float)ey;
        }
    for (int s = 0; s < 5; s++) {
        m->start[0][s] = s == 0 ? 1.0f : 0.0f;
        m->start[1][s] = (s == 3 || s == 4) ? 1.0f : 0.0f;
        m->end[0][s] = m->T[s][0];
    }
    m->end[1][0] = m->T[0][3];
    m->end[1][1] = m->T[0][3];
    m->end[1][2] = m->T[0][4];
    m->end[1][3] = m->T[3][3];
    m->end[1][4] = m->T[4][4];
}

static inline float from_bits(uint32_t b) {
    float f;
    memcpy(&f, &b, 4);
    return f;
}
static inline uint32_t to_bits(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    return b;
}

static const rcell ZERO = {{0.f, 0.f, 0.f, 0.f, 0.f}};

/* L = (x-1,y), M = (x-1,y-1), U = (x,y-1); cx = X[x-1], cy = Y[y-1] */
static inline void fwd_cell(rcell *c, const rmodel *m, const rcell *L, const rcell *M, const rcell *U, int cx, int cy) {
    const float(*T)[5] = m->T;
    float a;
    a = T[0][0] * M->v[0];
    a = fmaf(T[1][0], M->v[1], a);
    a = fmaf(T[2][0], M->v[2], a);
    a = fmaf(T[3][0], M->v[3], a);
    a = fmaf(T[4][0], M->v[4], a);
    c->v[0] = m->em[cx][cy] * a;
    a = T[0][1] * L->v[0];
    a = fmaf(T[1][1], L->v[1], a);
    a = fmaf(T[2][1], L->v[2], a);
    c->v[1] = m->ex[1][cx] * a;
    a = T[0][3] * L->v[0];
    a = fmaf(T[3][3], L->v[3], a);
    c->v[3] = m->ex[3][cx] * a;
    a = T[0][2] * U->v[0];
    a = fmaf(T[2][2], U->v[2], a);
    a = fmaf(T[1][2], U->v[1], a);
    c->v[2] = m->ey[2][cy] * a;
    a = T[0][4] * U->v[0];
    a = fmaf(T[4][4], U->v[4], a);
    c->v[4] = m->ey[4][cy] * a;
}

/* Ms = (x+1,y+1), Xs = (x+1,y), Ys = (x,y+1); cx = X[x], cy = Y[y] (the bases those moves consume) */
static inline void bwd_cell(rcell *c, const rmodel *m, const rcell *Ms, const rcell *Xs, const rcell *Ys, int cx, int cy) {
    const float(*T)[5] = m->T;
    const float am = m->em[cx][cy] * Ms->v[0];
    const float asx = m->ex[1][cx] * Xs->v[1];
    const float alx = m->ex[3][cx] * Xs->v[3];
    const float asy = m->ey[2][cy] * Ys->v[2];
    const float aly = m->ey[4][cy] * Ys->v[4];
    float b;
    b = T[0][0] * am;
    b = fmaf(T[0][1], asx, b);
    b = fmaf(T[0][3], alx, b);
    b = fmaf(T[0][2], asy, b);
    b = fmaf(T[0][4], aly, b);
    c->v[0] = b;
    b = T[1][0] * am;
    b = fmaf(T[1][1], asx, b);
    b = fmaf(T[1][2], asy, b);
    c->v[1] = b;
    b = T[2][0] * am;
    b = fmaf(T[2][2], asy, b);
    b = fmaf(T[2][1], asx, b);
    c->v[2] = b;
    b = T[3][0] * am;
    b = fmaf(T[3][3], alx, b);
    c->v[3] = b;
    b = T[4][0] * am;
    b = fmaf(T[4][4], aly, b);
    c->v[4] = b;
}

static inline float dot5(const float *w, const float *v) {
    float a = w[0] * v[0];
    a = fmaf(w[1], v[1], a);
    a = fmaf(w[2], v[2], a);
    a = fmaf(w[3], v[3], a);
    a = fmaf(w[4], v[4], a);
    return a;
}

/* The two held rows [a0, a1) and [b0, b1) of V: the largest bit pattern (the values are non-negative) decides the
 * power of two; returns what is added to the rows' exponent. */
static int32_t renorm(rcell *V, int64_t a0, int64_t a1, int64_t b0, int64_t b1) {
    uint32_t top = 0;
    for (int64_t i = a0; i < a1; i++)
        for (int s = 0; s < 5; s++)
            if (to_bits(V[i].v[s]) > top) top = to_bits(V[i].v[s]);
    for (int64_t i = b0; i < b1; i++)
        for (int s = 0; s < 5; s++)
            if (to_bits(V[i].v[s]) > top) top = to_bits(V[i].v[s]);
    const int32_t eb = (int32_t)(top >> 23);
    if (eb == 0) return 0;
    int32_t k = RS_TOP + 126 - eb; /* the maximum goes to [2^(RS_TOP-1), 2^RS_TOP) */
    if (k < -126) k = -126;
    if (k > 127) k = 127;
    const float f = from_bits((uint32_t)(k + 127) << 23); /* 2^k */
    for (int64_t i = a0; i < a1; i++)
        for (int s = 0; s < 5; s++) V[i].v[s] = V[i].v[s] * f;
    for (int64_t i = b0; i < b1; i++)
        for (int s = 0; s < 5; s++) V[i].v[s] = V[i].v[s] * f;
    return -k;
}

static inline int64_t cidx(const int32_t *lo, const int32_t *n, const int64_t *off, int64_t D, int64_t d, int64_t xmy) {
    if (d < 0 || d > D) return -1;
    int64_t j = xmy - lo[d];
    if (j < 0 || (j & 1)) return -1;
    j >>= 1;
    if (j >= n[d]) return -1;
    return off[d] + j;
}

static int model_supported(const orc_hmm *h) {
    static const int used[5][5] = {
        {1, 1, 1, 1, 1}, {1, 1, 1, 0, 0}, {1, 1, 1, 0, 0}, {1, 0, 0, 1, 0}, {1, 0, 0, 0, 1}};
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++)
            if (!used[a][b] && h->T[a * 5 + b] != 0.0) return 0;
    return 1;
}

int32_t orc_fb_f32_rs(const orc_hmm *h, const uint8_t *X, int64_t lX, const uint8_t *Y, int64_t lY, const int32_t *lo,
                      const int32_t *n, int32_t ragged_start, int32_t ragged_end, float threshold, float *tot_m,
                      int32_t *tot_e, float *btot_m, int32_t *btot_e, float *Fm_v, int32_t *Fm_e, float *Bm_v,
                      int32_t *Bm_e, int32_t *px, int32_t *py, float *pp, int64_t cap, int64_t *npairs) {
    if (!model_supported(h)) return -4;
    rmodel m;
    rmodel_init(&m, h);
    const int64_t D = lX + lY;
    int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(D + 2));
    off[0] = 0;
    for (int64_t d = 0; d <= D; d++) off[d + 1] = off[d] + n[d];
    const int64_t cells = off[D + 1];
    rcell *V = (rcell *)malloc(sizeof(rcell) * (size_t)(cells + 1));   /* the rows as the registers hold them */
    float *Fst = (float *)malloc(sizeof(float) * (size_t)(cells + 1)); /* match values as the rows went to the scratch */
    int32_t *Fex = (int32_t *)malloc(sizeof(int32_t) * (size_t)(D + 1));
    int32_t rc = 0;

    int32_t e = 0;
    for (int64_t d = 0; d <= D; d++) {
        for (int64_t j = 0; j < n[d]; j++) {
            const int64_t xmy = lo[d] + 2 * j, x = (d + xmy) / 2, y = (d - xmy) / 2;
            rcell *c = V + off[d] + j;
            if (x < 0 || y < 0 || x > lX || y > lY) {
                *c = ZERO;
            } else if (d == 0) {
                for (int s = 0; s < 5; s++) c->v[s] = m.start[ragged_start ? 1 : 0][s];
            } else {
                const int64_t iM = (x > 0 && y > 0) ? cidx(lo, n, off, D, d - 2, xmy) : -1;
                const int64_t iL = (x > 0) ? cidx(lo, n, off, D, d - 1, xmy - 1) : -1;
                const int64_t iU = (y > 0) ? cidx(lo, n, off, D, d - 1, xmy + 1) : -1;
                fwd_cell(c, &m, iL >= 0 ? V + iL : &ZERO, iM >= 0 ? V + iM : &ZERO, iU >= 0 ? V + iU : &ZERO,
                         x > 0 ? X[x - 1] : 4, y > 0 ? Y[y - 1] : 4);
            }
        }
        if (d > 0 && d % RS_K == 0) e += renorm(V, off[d], off[d + 1], off[d - 1], off[d]);
        for (int64_t i = off[d]; i < off[d + 1]; i++) Fst[i] = V[i].v[0];
        Fex[d] = e;
    }
    float tm = 0.f;
    int32_t te = E_DEAD;
    {
        const int64_t ie = cidx(lo, n, off, D, D, lX - lY);
        if (ie >= 0) {
            float raw = dot5(m.end[ragged_end ? 1 : 0], V[ie].v);
            if (raw > 0.0f) {
                int k;
                tm = frexpf(raw, &k);
                te = e + k;
            }
        }
    }
    if (tot_m) *tot_m = tm;
    if (tot_e) *tot_e = te;
    if (!(tm > 0.0f)) rc = -2;
    if (Fm_v)
        for (int64_t d = 0; d <= D; d++)
            for (int64_t i = off[d]; i < off[d + 1]; i++) Fm_v[i] = Fst[i], Fm_e[i] = Fex[d];

    int64_t np = 0;
    if (rc == 0) {
        const float inv_tot = 1.0f / tm;
        int32_t eb = 0;
        int32_t smax = -(1 << 30);
        for (int64_t d = D; d >= 0; d--) {
            for (int64_t j = 0; j < n[d]; j++) {
                const int64_t xmy = lo[d] + 2 * j, x = (d + xmy) / 2, y = (d - xmy) / 2;
                rcell *c = V + off[d] + j;
                if (x < 0 || y < 0 || x > lX || y > lY) {
                    *c = ZERO;
                } else if (d == D) {
                    for (int s = 0; s < 5; s++) c->v[s] = m.end[ragged_end ? 1 : 0][s];
                } else {
                    const int64_t jM = (x < lX && y < lY) ? cidx(lo, n, off, D, d + 2, xmy) : -1;
                    const int64_t jX = (x < lX) ? cidx(lo, n, off, D, d + 1, xmy + 1) : -1;
                    const int64_t jY = (y < lY) ? cidx(lo, n, off, D, d + 1, xmy - 1) : -1;
                    bwd_cell(c, &m, jM >= 0 ? V + jM : &ZERO, jX >= 0 ? V + jX : &ZERO, jY >= 0 ? V + jY : &ZERO,
                             x < lX ? X[x] : 4, y < lY ? Y[y] : 4);
                }
            }
            if (d < D && d % RS_K == 0) eb += renorm(V, off[d], off[d + 1], off[d + 1], off[d + 2]);
            if (Bm_v)
                for (int64_t i = off[d]; i < off[d + 1]; i++) Bm_v[i] = V[i].v[0], Bm_e[i] = eb;
            /* posterior of the row: p = (Fv * (Bv * 2^(eF+eB-eTot))) * (1/totMant) */
            {
                const int32_t s = Fex[d] + eb - te;
                if (s > smax) smax = s;
            }
            if (d >= 2 && (px || npairs)) {
                const int32_t s = Fex[d] + eb - te;
                for (int64_t j = 0; j < n[d]; j++) {
                    const int64_t xmy = lo[d] + 2 * j, x = (d + xmy) / 2, y = (d - xmy) / 2;
                    if (x < 1 || y < 1 || x > lX || y > lY) continue;
                    const int64_t ic = off[d] + j;
                    const float pr = (Fst[ic] * ldexpf(V[ic].v[0], s)) * inv_tot;
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
        if (btot_m) {
            float raw = dot5(m.start[ragged_start ? 1 : 0], V[0].v);
            int k = 0;
            *btot_m = raw > 0.0f ? frexpf(raw, &k) : 0.0f;
            *btot_e = raw > 0.0f ? eb + k : E_DEAD;
        }
        /* the range certificate failed: the device runs such a task again in the per-cell arithmetic (results stay valid
         * as a restatement of what the row-scaled kernel computed) */
        if (rc == 0 && smax >= RS_S_LIMIT) rc = 1;
    }
    if (npairs) *npairs = np;
    free(V);
    free(Fst);
    free(Fex);
    free(off);
    return rc;
}
