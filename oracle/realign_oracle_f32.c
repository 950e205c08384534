/*
 * realign_oracle_f32.c -- CPU ORACLE, fp32 "mirror" of the device arithmetic.
 * TEST INFRASTRUCTURE ONLY (see realign_oracle.h).
 *
 * The HIP kernels evaluate the same forward/backward recurrences as orc_fb_f64 (SURVEY 8a rows
 * a5.3-a5.5; call sites nanopore/analyses/utils.py:587, alignmentUncertainty.py:41,
 * marginAlignSnpCaller.py:136-146) but represent every DP cell in block floating point: five linear
 * fp32 mantissas that share one int32 binary exponent, i.e. log2(value) = e + log2(v).  This is the
 * log-sum-exp recurrence with the integer part of the logarithm carried exactly and the fractional
 * part carried linearly, so no exp/log is needed inside the recurrence and every operation is an
 * IEEE-754 single-rounding op (mul, fma, ldexp, frexp, max) that a CPU reproduces bit for bit.
 * This file restates that arithmetic (DESIGN.md "Device arithmetic") operation by operation; the
 * parity tests require the GPU results to be IDENTICAL to it, and require it to agree with the
 * double-precision log-space oracle within the stated tolerance.
 *
 * Compile with -ffp-contract=off: every fused multiply-add below is an explicit fmaf().
 */
#include "realign_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define E_DEAD (-(1 << 28))

typedef struct {
    float v[5]; /* 0 match, 1 shortGapX, 2 shortGapY, 3 longGapX, 4 longGapY */
    int32_t e;
} cell32;

typedef struct {
    float T[5][5];
    float em[5][5];
    float ex[5][5]; /* [state][x] */
    float ey[5][5]; /* [state][y] */
    float start[2][5], end[2][5];
} model32;

static void model32_init(model32 *m, const orc_hmm *h) {
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

static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

static inline float from_bits(int32_t b) {
    float f;
    memcpy(&f, &b, 4);
    return f;
}
static inline int32_t to_bits(float f) {
    int32_t b;
    memcpy(&b, &f, 4);
    return b;
}

/* 2^k for -126 <= k <= 0 from its bit pattern, 0 below (k is an exponent difference, never positive) */
static inline float scale2(int32_t k) {
    const int32_t t = k + 127;
    return from_bits((t > 0 ? t : 0) << 23);
}

/* multiply by 2^(126 - E), E the biased exponent field of the largest value; shared exponent += E - 126 */
static inline void normalise(cell32 *c, int32_t eref) {
    const float vmax = fmaxf(fmaxf(c->v[0], c->v[1]), fmaxf(fmaxf(c->v[2], c->v[3]), c->v[4]));
    const int32_t bits = to_bits(vmax) & 0x7f800000;
    const float inv = from_bits(0x7e800000 - bits);
    for (int s = 0; s < 5; s++) c->v[s] = c->v[s] * inv;
    c->e = vmax > 0.0f ? eref + (bits >> 23) - 126 : E_DEAD;
}

static const cell32 DEAD = {{0.f, 0.f, 0.f, 0.f, 0.f}, E_DEAD};

/* Which anti-diagonals renormalise (npr_cell.h norm_diag): d = 0, 1 (mod 4).  On the others a cell keeps the
 * reference exponent of its predecessors and the mantissas the recurrence produced. */
static inline int norm_diag(int64_t d) { return (d & 2) == 0; }
static inline void settle(cell32 *c, int32_t eref, int norm) {
    if (norm) normalise(c, eref);
    else c->e = eref;
}

static inline void fwd_cell(cell32 *c, const model32 *m, const cell32 *L, const cell32 *M, const cell32 *U, int cx,
                            int cy, int norm) {
    const int32_t eref = imax(L->e, imax(M->e, U->e));
    const float fL = scale2(L->e - eref), fM = scale2(M->e - eref), fU = scale2(U->e - eref);
    const float(*T)[5] = m->T;
    float a;
    a = T[0][0] * M->v[0];
    a = fmaf(T[1][0], M->v[1], a);
    a = fmaf(T[2][0], M->v[2], a);
    a = fmaf(T[3][0], M->v[3], a);
    a = fmaf(T[4][0], M->v[4], a);
    c->v[0] = (fM * m->em[cx][cy]) * a;
    a = T[0][1] * L->v[0];
    a = fmaf(T[1][1], L->v[1], a);
    a = fmaf(T[2][1], L->v[2], a);
    c->v[1] = (fL * m->ex[1][cx]) * a;
    a = T[0][3] * L->v[0];
    a = fmaf(T[3][3], L->v[3], a);
    c->v[3] = (fL * m->ex[3][cx]) * a;
    a = T[0][2] * U->v[0];
    a = fmaf(T[2][2], U->v[2], a);
    a = fmaf(T[1][2], U->v[1], a);
    c->v[2] = (fU * m->ey[2][cy]) * a;
    a = T[0][4] * U->v[0];
    a = fmaf(T[4][4], U->v[4], a);
    c->v[4] = (fU * m->ey[4][cy]) * a;
    settle(c, eref, norm);
}

/* Ms = (x+1,y+1), Xs = (x+1,y), Ys = (x,y+1); cx = X[x], cy = Y[y] (the bases those moves consume) */
static inline void bwd_cell(cell32 *c, const model32 *m, const cell32 *Ms, const cell32 *Xs, const cell32 *Ys, int cx,
                            int cy, int norm) {
    const int32_t eref = imax(Ms->e, imax(Xs->e, Ys->e));
    const float fM = scale2(Ms->e - eref), fX = scale2(Xs->e - eref), fY = scale2(Ys->e - eref);
    const float(*T)[5] = m->T;
    const float am = (fM * m->em[cx][cy]) * Ms->v[0];
    const float asx = (fX * m->ex[1][cx]) * Xs->v[1];
    const float alx = (fX * m->ex[3][cx]) * Xs->v[3];
    const float asy = (fY * m->ey[2][cy]) * Ys->v[2];
    const float aly = (fY * m->ey[4][cy]) * Ys->v[4];
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
    settle(c, eref, norm);
}

static inline float dot5(const float *w, const float *v) {
    float a = w[0] * v[0];
    a = fmaf(w[1], v[1], a);
    a = fmaf(w[2], v[2], a);
    a = fmaf(w[3], v[3], a);
    a = fmaf(w[4], v[4], a);
    return a;
}

static inline int64_t cidx(const int32_t *lo, const int32_t *n, const int64_t *off, int64_t D, int64_t d, int64_t xmy) {
    if (d < 0 || d > D) return -1;
    int64_t j = xmy - lo[d];
    if (j < 0 || (j & 1)) return -1;
    j >>= 1;
    if (j >= n[d]) return -1;
    return off[d] + j;
}

/* the 15 transitions the device kernels evaluate (cPecan's five-state cell update); a model with any
 * other non-zero transition is rejected by the product with NPR_ERR_MODEL and is rejected here too */
static int model_supported(const orc_hmm *h) {
    static const int used[5][5] = {
        {1, 1, 1, 1, 1}, {1, 1, 1, 0, 0}, {1, 1, 1, 0, 0}, {1, 0, 0, 1, 0}, {1, 0, 0, 0, 1}};
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++)
            if (!used[a][b] && h->T[a * 5 + b] != 0.0) return 0;
    return 1;
}

int32_t orc_fb_f32(const orc_hmm *h, const uint8_t *X, int64_t lX, const uint8_t *Y, int64_t lY,
                   const int32_t *lo, const int32_t *n, int32_t ragged_start, int32_t ragged_end,
                   float threshold, float *tot_m, int32_t *tot_e, float *btot_m, int32_t *btot_e,
                   float *Fm_v, int32_t *Fm_e, float *Bm_v, int32_t *Bm_e, int32_t *px, int32_t *py,
                   float *pp, int64_t cap, int64_t *npairs) {
    if (!model_supported(h)) return -4;
    model32 m;
    model32_init(&m, h);
    const int64_t D = lX + lY;
    int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(D + 2));
    off[0] = 0;
    for (int64_t d = 0; d <= D; d++) off[d + 1] = off[d] + n[d];
    const int64_t cells = off[D + 1];
    cell32 *F = (cell32 *)malloc(sizeof(cell32) * (size_t)cells);
    cell32 *B = (cell32 *)malloc(sizeof(cell32) * (size_t)cells);
    int32_t rc = 0;

    for (int64_t d = 0; d <= D; d++)
        for (int64_t j = 0; j < n[d]; j++) {
            const int64_t xmy = lo[d] + 2 * j, x = (d + xmy) / 2, y = (d - xmy) / 2;
            cell32 *c = F + off[d] + j;
            if (x < 0 || y < 0 || x > lX || y > lY) {
                *c = DEAD;
            } else if (d == 0) {
                for (int s = 0; s < 5; s++) c->v[s] = m.start[ragged_start ? 1 : 0][s];
                normalise(c, 0);
            } else {
                const int64_t iM = (x > 0 && y > 0) ? cidx(lo, n, off, D, d - 2, xmy) : -1;
                const int64_t iL = (x > 0) ? cidx(lo, n, off, D, d - 1, xmy - 1) : -1;
                const int64_t iU = (y > 0) ? cidx(lo, n, off, D, d - 1, xmy + 1) : -1;
                fwd_cell(c, &m, iL >= 0 ? F + iL : &DEAD, iM >= 0 ? F + iM : &DEAD, iU >= 0 ? F + iU : &DEAD,
                         x > 0 ? X[x - 1] : 4, y > 0 ? Y[y - 1] : 4, norm_diag(d));
            }
        }
    float tm = 0.f;
    int32_t te = E_DEAD;
    {
        const int64_t ie = cidx(lo, n, off, D, D, lX - lY);
        if (ie >= 0) {
            float raw = dot5(m.end[ragged_end ? 1 : 0], F[ie].v);
            if (raw > 0.0f) {
                int k;
                tm = frexpf(raw, &k);
                te = F[ie].e + k;
            }
        }
    }
    if (tot_m) *tot_m = tm;
    if (tot_e) *tot_e = te;
    if (!(tm > 0.0f)) rc = -2;
    if (Fm_v)
        for (int64_t i = 0; i < cells; i++) Fm_v[i] = F[i].v[0], Fm_e[i] = F[i].e;

    int64_t np = 0;
    if (rc == 0) {
        for (int64_t d = D; d >= 0; d--)
            for (int64_t j = 0; j < n[d]; j++) {
                const int64_t xmy = lo[d] + 2 * j, x = (d + xmy) / 2, y = (d - xmy) / 2;
                cell32 *c = B + off[d] + j;
                if (x < 0 || y < 0 || x > lX || y > lY) {
                    *c = DEAD;
                } else if (d == D) {
                    for (int s = 0; s < 5; s++) c->v[s] = m.end[ragged_end ? 1 : 0][s];
                    normalise(c, 0);
                } else {
                    const int64_t jM = (x < lX && y < lY) ? cidx(lo, n, off, D, d + 2, xmy) : -1;
                    const int64_t jX = (x < lX) ? cidx(lo, n, off, D, d + 1, xmy + 1) : -1;
                    const int64_t jY = (y < lY) ? cidx(lo, n, off, D, d + 1, xmy - 1) : -1;
                    bwd_cell(c, &m, jM >= 0 ? B + jM : &DEAD, jX >= 0 ? B + jX : &DEAD, jY >= 0 ? B + jY : &DEAD,
                             x < lX ? X[x] : 4, y < lY ? Y[y] : 4, norm_diag(d));
                }
            }
        if (btot_m) {
            float raw = dot5(m.start[ragged_start ? 1 : 0], B[0].v);
            int k = 0;
            *btot_m = raw > 0.0f ? frexpf(raw, &k) : 0.0f;
            *btot_e = raw > 0.0f ? B[0].e + k : E_DEAD;
        }
        if (Bm_v)
            for (int64_t i = 0; i < cells; i++) Bm_v[i] = B[i].v[0], Bm_e[i] = B[i].e;
        /* posterior: p = (Fv*Bv) * 2^(eF+eB-eTot) * (1/totMant), diagonal ascending, xmy ascending */
        const float inv_tot = 1.0f / tm;
        for (int64_t d = 2; d <= D && (px || npairs); d++)
            for (int64_t j = 0; j < n[d]; j++) {
                const int64_t xmy = lo[d] + 2 * j, x = (d + xmy) / 2, y = (d - xmy) / 2;
                if (x < 1 || y < 1 || x > lX || y > lY) continue;
                const int64_t ic = off[d] + j;
                if (F[ic].e == E_DEAD || B[ic].e == E_DEAD) continue;
                int32_t s = F[ic].e + B[ic].e - te;
                if (s < -200) s = -200;
                if (s > 200) s = 200;
                const float q = F[ic].v[0] * B[ic].v[0];
                const float pr = ldexpf(q, s) * inv_tot;
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
    if (npairs) *npairs = np;
    free(F);
    free(B);
    free(off);
    return rc;
}
