/*
 * llz_iir_host.c -- handle layer of the IIR path: the reference's single-channel direct-form-I symbols
 * (reference libllzfilter/llz_iir.c:37-156) and the multi-channel biquad-cascade extension.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../../include/llz_iir.h"
#include "llz_host.h"

/* ---- Part 1: single channel, double, exact order (k_iir_df1_f64_exact) ---- */

typedef struct {
    int tag;
    int M, N;
    double *d_a, *d_b, *d_xs, *d_ys;     /* coefficients and delay lines live on the device between calls */
    llz_stage_t st_in, st_out;
} iir1_t;

static void iir1_destroy(iir1_t *f)
{
    if (!f) return;
    llzs_free(f->d_a); llzs_free(f->d_b); llzs_free(f->d_xs); llzs_free(f->d_ys);
    llz_stage_release(&f->st_in); llz_stage_release(&f->st_out);
    f->tag = 0;
    free(f);
}

unsigned long llz_iir_filter_init(int M, double *a, int N, double *b)
{
    if (M < 0 || N < 0 || M > 1024 || N > 1024 || !a) {
        llzs_set_error("llz_iir_filter_init: M=%d N=%d (0..1024) or NULL a", M, N);
        return LLZ_BAD_HANDLE;
    }
    iir1_t *f = (iir1_t *)calloc(1, sizeof(*f));
    if (!f) return LLZ_BAD_HANDLE;
    f->tag = LLZ_TAG_IIR1;
    f->M = M; f->N = N;
    double *bz = (double *)calloc((size_t)N + 1, sizeof(double));      /* b == NULL -> zeros (llz_iir.c:54-59) */
    int rc = bz ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK) {
        if (b) memcpy(bz, b, sizeof(double) * ((size_t)N + 1));
        f->d_a = (double *)llzs_malloc(sizeof(double) * ((size_t)M + 1));
        f->d_b = (double *)llzs_malloc(sizeof(double) * ((size_t)N + 1));
        f->d_xs = (double *)llzs_malloc(sizeof(double) * ((size_t)N + 1));
        f->d_ys = (double *)llzs_malloc(sizeof(double) * ((size_t)M + 1));
        if (!f->d_a || !f->d_b || !f->d_xs || !f->d_ys) rc = LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK) rc = llzs_h2d(f->d_a, a, sizeof(double) * ((size_t)M + 1), NULL);
    if (rc == LLZ_OK) rc = llzs_h2d(f->d_b, bz, sizeof(double) * ((size_t)N + 1), NULL);
    if (rc == LLZ_OK) rc = llzs_memset(f->d_xs, 0, sizeof(double) * ((size_t)N + 1), NULL);
    if (rc == LLZ_OK) rc = llzs_memset(f->d_ys, 0, sizeof(double) * ((size_t)M + 1), NULL);
    if (rc == LLZ_OK) rc = llzs_sync(NULL);
    free(bz);
    if (rc != LLZ_OK) {
        iir1_destroy(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

void llz_iir_filter_uninit(unsigned long handle)
{
    if (LLZ_HANDLE_OK(handle, iir1_t, LLZ_TAG_IIR1))
        iir1_destroy((iir1_t *)handle);
}

static int iir1_run(iir1_t *f, const double *x, double *y, int n)
{
    const size_t bytes = sizeof(double) * (size_t)n;
    double *d_in = (double *)llz_stage_reserve(&f->st_in, bytes);
    double *d_out = (double *)llz_stage_reserve(&f->st_out, bytes);
    if (!d_in || !d_out) return LLZ_ERR_NOMEM;
    int rc = x ? llzs_h2d(d_in, x, bytes, NULL) : llzs_memset(d_in, 0, bytes, NULL);
    if (rc == LLZ_OK) rc = llzs_iir_df1_f64(d_in, d_out, f->d_a, f->d_b, f->d_xs, f->d_ys, f->M, f->N, n, NULL);
    if (rc == LLZ_OK) rc = llzs_d2h(y, d_out, bytes, NULL);
    return rc;
}

int llz_iir_filter(unsigned long handle, double *x, double *y, int frame_len)
{
    if (!LLZ_HANDLE_OK(handle, iir1_t, LLZ_TAG_IIR1) || !x || !y || frame_len < 0) {
        llzs_set_error("llz_iir_filter: bad handle or buffer");
        return LLZ_ERR_ARG;
    }
    if (frame_len == 0) return 0;
    const int rc = iir1_run((iir1_t *)handle, x, y, frame_len);
    return rc == LLZ_OK ? frame_len : rc;
}

int llz_iir_filter_flush(unsigned long handle, double *y)
{
    if (!LLZ_HANDLE_OK(handle, iir1_t, LLZ_TAG_IIR1) || !y) {
        llzs_set_error("llz_iir_filter_flush: bad handle or buffer");
        return LLZ_ERR_ARG;
    }
    iir1_t *f = (iir1_t *)handle;
    if (f->N == 0) return 0;
    const int rc = iir1_run(f, NULL, y, f->N);            /* N samples of x = 0 (llz_iir.c:147-156) */
    return rc == LLZ_OK ? f->N : rc;
}

/* ---- Part 2: multi-channel biquad cascade (k_iir_cascade_f32) ---- */

typedef struct {
    int tag;
    int channels, stages;
    double *d_coef;      /* stages x {b0,b1,b2,a1,a2} */
    double *d_pd, *d_pl; /* state-transition powers for the pipelined kernel: [S][6][4] and [S][64][12] */
    float *d_coef32, *d_pd32, *d_pl32;   /* float copies for the wave-autonomous float32 kernel: [S][5], [S][16], [S][64][12] */
    float *d_pd32w, *d_pl32w, *d_ph32w;   /* 32-sample-per-lane packed kernel: powers of A^32, ph [S][40] (b0 folded) */
    float in_gain32;     /* the product of the b0's (that kernel scales the input once) */
    double *d_cw64, *d_pd64w, *d_pl64w;   /* the same form in double (k_iir_cascade_wave_pf64w): cw [S][8], pd [S][16], plc [S][448] */
    double in_gain64;
    float *d_ph32;       /* [S][24]: (h1[k], h2[k]) k < 8 = zero-input outputs of the unit start states, b0 b1 b2 a1 a2, pad */
    double *d_state;     /* [channels][stages][x1,x2,y1,y2]: the current state */
    double *d_state_alt; /* where a time-segmented launch writes the frame's end state (then the two swap) */
    int warm_chunks;     /* 1024-sample chunks after which any state error has decayed below 1e-13 (0: unknown / too long) */
    int float32_ok;      /* every section's rounding-noise gain is small enough for float32 arithmetic */
    void *stream;
    int device;          /* the device the handle's buffers live on: every call binds it */
    llz_stage_t st_in, st_out;
} iirm_t;

static void iirm_destroy(iirm_t *f)
{
    if (!f) return;
    llzs_free(f->d_coef); llzs_free(f->d_state); llzs_free(f->d_state_alt); llzs_free(f->d_pd); llzs_free(f->d_pl);
    llzs_free(f->d_coef32); llzs_free(f->d_pd32); llzs_free(f->d_pl32); llzs_free(f->d_ph32);
    llzs_free(f->d_pd32w); llzs_free(f->d_pl32w); llzs_free(f->d_ph32w);
    llzs_free(f->d_cw64); llzs_free(f->d_pd64w); llzs_free(f->d_pl64w);
    llz_stage_release(&f->st_in); llz_stage_release(&f->st_out);
    f->tag = 0;
    free(f);
}

static void mat2_mul(const double *a, const double *b, double *o)
{
    const double r0 = a[0] * b[0] + a[1] * b[2], r1 = a[0] * b[1] + a[1] * b[3];
    const double r2 = a[2] * b[0] + a[3] * b[2], r3 = a[2] * b[1] + a[3] * b[3];
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}

/* transition matrices of the feedback recurrence (y[n-1], y[n-2]) -> 16 samples later, its powers of two for the
 * lane scan and its lane-th powers: see k_iir_cascade_pipe_f32 */
static int iirm_build_powers(iirm_t *f, const double *c5, int lane_run)
{
    const int S = f->stages;
    double *pd = (double *)malloc(sizeof(double) * (size_t)S * 6 * 4);
    double *pl = (double *)malloc(sizeof(double) * (size_t)S * 64 * 12);
    int rc = (pd && pl) ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK) {
        for (int s = 0; s < S; s++) {
            const double A[4] = {-c5[5 * s + 3], -c5[5 * s + 4], 1.0, 0.0};
            double P[4] = {1.0, 0.0, 0.0, 1.0};
            for (int i = 0; i < lane_run; i++) mat2_mul(A, P, P);      /* P = A^lane_run (16 or 32 samples per lane) */
            double *d = pd + (size_t)s * 24;
            memcpy(d, P, sizeof(P));
            for (int k = 1; k < 6; k++) mat2_mul(d + 4 * (k - 1), d + 4 * (k - 1), d + 4 * k);
            double pw[65][4];                                          /* P^0 .. P^64 */
            pw[0][0] = 1.0; pw[0][1] = 0.0; pw[0][2] = 0.0; pw[0][3] = 1.0;
            for (int k = 1; k <= 64; k++) mat2_mul(P, pw[k - 1], pw[k]);
            for (int lane = 0; lane < 64; lane++) {                    /* per lane: P^lane, P^(lane%16+1), P^(lane%32+1) */
                double *l = pl + ((size_t)s * 64 + lane) * 12;
                memcpy(l, pw[lane], sizeof(pw[0]));
                memcpy(l + 4, pw[lane % 16 + 1], sizeof(pw[0]));
                memcpy(l + 8, pw[lane % 32 + 1], sizeof(pw[0]));
            }
        }
        f->d_pd = (double *)llzs_malloc(sizeof(double) * (size_t)S * 24);
        f->d_pl = (double *)llzs_malloc(sizeof(double) * (size_t)S * 768);
        rc = (f->d_pd && f->d_pl) ? LLZ_OK : LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_pd, pd, sizeof(double) * (size_t)S * 24);
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_pl, pl, sizeof(double) * (size_t)S * 768);
        if (rc == LLZ_OK && f->float32_ok) {                           /* float copies for k_iir_cascade_wave_f32 */
            float *t = (float *)malloc(sizeof(float) * (size_t)S * (5 + 16 + 768 + 24));
            if (!t) rc = LLZ_ERR_NOMEM;
            if (rc == LLZ_OK) {
                float *c32 = t, *pd32 = t + 5 * S, *pl32 = pd32 + 16 * S;
                for (int i = 0; i < 5 * S; i++) c32[i] = (float)c5[i];
                for (int s = 0; s < S; s++)
                    for (int i = 0; i < 16; i++) pd32[16 * s + i] = (float)pd[24 * s + i];
                for (int i = 0; i < 768 * S; i++) pl32[i] = (float)pl[i];
                float *ph32 = pl32 + 768 * S;
                for (int s = 0; s < S; s++) {                          /* y[k] from (y[-1], y[-2]) = (1,0) and (0,1) */
                    const double a1 = c5[5 * s + 3], a2 = c5[5 * s + 4];
                    double p1 = 1.0, p2 = 0.0, q1 = 0.0, q2 = 1.0;
                    for (int k = 0; k < 8; k++) {
                        const double h1 = -a1 * p1 - a2 * p2, h2 = -a1 * q1 - a2 * q2;
                        ph32[24 * s + 2 * k] = (float)h1; ph32[24 * s + 2 * k + 1] = (float)h2;
                        p2 = p1; p1 = h1; q2 = q1; q1 = h2;
                    }
                    for (int k = 0; k < 8; k++) ph32[24 * s + 16 + k] = k < 5 ? (float)c5[5 * s + k] : 0.f;
                }
                f->d_coef32 = (float *)llzs_malloc(sizeof(float) * 5 * (size_t)S);
                f->d_pd32 = (float *)llzs_malloc(sizeof(float) * 16 * (size_t)S);
                f->d_pl32 = (float *)llzs_malloc(sizeof(float) * 768 * (size_t)S);
                f->d_ph32 = (float *)llzs_malloc(sizeof(float) * 24 * (size_t)S);
                if (!f->d_coef32 || !f->d_pd32 || !f->d_pl32 || !f->d_ph32) rc = LLZ_ERR_NOMEM;
                if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_coef32, c32, sizeof(float) * 5 * (size_t)S);
                if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_pd32, pd32, sizeof(float) * 16 * (size_t)S);
                if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_pl32, pl32, sizeof(float) * 768 * (size_t)S);
                if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_ph32, ph32, sizeof(float) * 24 * (size_t)S);
            }
            free(t);
        }
    }
    free(pd); free(pl);
    return rc;
}

/* Tables of the packed float32 kernel with 32 samples per lane and the b0 gains folded out (k_iir_cascade_wave_pk32):
 * powers of P = A^32 for the lane scan, the homogeneous responses h1[k], h2[k] for k < 16, the section's b1/b0, b2/b0,
 * a1, a2, and the state scales xfac_s = prod_{t >= s} b0_t, yfac_s = prod_{t > s} b0_t.  Built only when every b0 is
 * usable as a divisor (no zero gain, partial products between 1e-20 and 1e20); otherwise the 16-sample kernel runs. */
static int iirm_build_run32(iirm_t *f, const double *c5)
{
    const int S = f->stages;
    if (S > 8 || !f->float32_ok) return LLZ_OK;
    double xfac[9];
    xfac[S] = 1.0;
    for (int s = S - 1; s >= 0; s--) {
        const double b0 = c5[5 * s];
        xfac[s] = xfac[s + 1] * b0;
        /* (the scaled states and the scaled input must stay clear of the float32 denormals: signals down to 1e-10 of full
         *  scale times a partial product of 1e-20 are still 1e-30) */
        if (!(fabs(b0) > 1e-12) || !(fabs(xfac[s]) > 1e-20 && fabs(xfac[s]) < 1e20) ||
            !(fabs(c5[5 * s + 1] / b0) < 1e4) || !(fabs(c5[5 * s + 2] / b0) < 1e4)) return LLZ_OK;
    }
    float *t = (float *)calloc((size_t)S * (16 + 768 + 40), sizeof(float));
    if (!t) return LLZ_ERR_NOMEM;
    float *pd = t, *pl = t + 16 * S, *ph = pl + 768 * S;
    for (int s = 0; s < S; s++) {
        const double a1 = c5[5 * s + 3], a2 = c5[5 * s + 4];
        const double A[4] = {-a1, -a2, 1.0, 0.0};
        double P[4] = {1.0, 0.0, 0.0, 1.0};
        for (int i = 0; i < 32; i++) mat2_mul(A, P, P);
        double pw2[4][4];                                          /* P^(2^d), d < 4 */
        memcpy(pw2[0], P, sizeof(P));
        for (int d = 1; d < 4; d++) mat2_mul(pw2[d - 1], pw2[d - 1], pw2[d]);
        for (int d = 0; d < 4; d++)
            for (int i = 0; i < 4; i++) pd[16 * s + 4 * d + i] = (float)pw2[d][i];
        double pw[65][4];
        pw[0][0] = 1.0; pw[0][1] = 0.0; pw[0][2] = 0.0; pw[0][3] = 1.0;
        for (int k = 1; k <= 64; k++) mat2_mul(P, pw[k - 1], pw[k]);
        for (int lane = 0; lane < 64; lane++) {                    /* per lane: P^lane, P^(lane%16+1), P^(lane%32+1) */
            float *l = pl + ((size_t)s * 64 + lane) * 12;
            for (int i = 0; i < 4; i++) {
                l[i] = (float)pw[lane][i]; l[4 + i] = (float)pw[lane % 16 + 1][i]; l[8 + i] = (float)pw[lane % 32 + 1][i];
            }
        }
        double p1 = 1.0, p2 = 0.0, q1 = 0.0, q2 = 1.0;             /* y[k] from (y[-1], y[-2]) = (1,0) and (0,1) */
        for (int k = 0; k < 16; k++) {
            const double h1 = -a1 * p1 - a2 * p2, h2 = -a1 * q1 - a2 * q2;
            ph[40 * s + 2 * k] = (float)h1; ph[40 * s + 2 * k + 1] = (float)h2;
            p2 = p1; p1 = h1; q2 = q1; q1 = h2;
        }
        const double b0 = c5[5 * s];
        ph[40 * s + 32] = 1.f;
        ph[40 * s + 33] = (float)(c5[5 * s + 1] / b0);
        ph[40 * s + 34] = (float)(c5[5 * s + 2] / b0);
        ph[40 * s + 35] = (float)a1;
        ph[40 * s + 36] = (float)a2;
        ph[40 * s + 37] = (float)xfac[s];
        ph[40 * s + 38] = (float)xfac[s + 1];
    }
    f->in_gain32 = (float)xfac[0];
    f->d_pd32w = (float *)llzs_malloc(sizeof(float) * 16 * (size_t)S);
    f->d_pl32w = (float *)llzs_malloc(sizeof(float) * 768 * (size_t)S);
    f->d_ph32w = (float *)llzs_malloc(sizeof(float) * 40 * (size_t)S);
    int rc = (f->d_pd32w && f->d_pl32w && f->d_ph32w) ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_pd32w, pd, sizeof(float) * 16 * (size_t)S);
    if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_pl32w, pl, sizeof(float) * 768 * (size_t)S);
    if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_ph32w, ph, sizeof(float) * 40 * (size_t)S);
    free(t);
    return rc;
}

/* Tables of the double kernel with 32 samples per lane and the b0 gains folded out (k_iir_cascade_wave_pf64w), for the
 * cascades float32 arithmetic is not good enough for: cw [S][8] = b1/b0, b2/b0, a1, a2, xfac_s, xfac_(s+1), 0, 0 with
 * xfac_s = prod_{t >= s} b0_t; pd [S][16] = P^(2^d), d < 4, P = A^32; plc [S][448] = P^lane for 64 lanes, P^(i+1) for
 * i < 16, P^(i+1) for i < 32 (2 x 2, row major).  The b0.s must be usable as divisors (partial products 1e-150..1e150). */
static int iirm_build_run32d(iirm_t *f, const double *c5)
{
    const int S = f->stages;
    if (S > 8 || f->float32_ok) return LLZ_OK;
    double xfac[9];
    xfac[S] = 1.0;
    for (int s = S - 1; s >= 0; s--) {
        const double b0 = c5[5 * s];
        xfac[s] = xfac[s + 1] * b0;
        if (!(fabs(b0) > 1e-30) || !(fabs(xfac[s]) > 1e-150 && fabs(xfac[s]) < 1e150) ||
            !(fabs(c5[5 * s + 1] / b0) < 1e6) || !(fabs(c5[5 * s + 2] / b0) < 1e6)) return LLZ_OK;
    }
    double *t = (double *)calloc((size_t)S * (8 + 16 + 448), sizeof(double));
    if (!t) return LLZ_ERR_NOMEM;
    double *cw = t, *pd = t + 8 * S, *pl = pd + 16 * S;
    for (int s = 0; s < S; s++) {
        const double b0 = c5[5 * s], a1 = c5[5 * s + 3], a2 = c5[5 * s + 4];
        const double A[4] = {-a1, -a2, 1.0, 0.0};
        double P[4] = {1.0, 0.0, 0.0, 1.0};
        for (int i = 0; i < 32; i++) mat2_mul(A, P, P);
        memcpy(pd + 16 * s, P, sizeof(P));
        for (int d = 1; d < 4; d++) mat2_mul(pd + 16 * s + 4 * (d - 1), pd + 16 * s + 4 * (d - 1), pd + 16 * s + 4 * d);
        double pw[65][4];
        pw[0][0] = 1.0; pw[0][1] = 0.0; pw[0][2] = 0.0; pw[0][3] = 1.0;
        for (int k = 1; k <= 64; k++) mat2_mul(P, pw[k - 1], pw[k]);
        double *l = pl + (size_t)s * 448;
        for (int lane = 0; lane < 64; lane++) memcpy(l + 4 * lane, pw[lane], sizeof(pw[0]));
        for (int i = 0; i < 16; i++) memcpy(l + 256 + 4 * i, pw[i + 1], sizeof(pw[0]));
        for (int i = 0; i < 32; i++) memcpy(l + 320 + 4 * i, pw[i + 1], sizeof(pw[0]));
        cw[8 * s + 0] = c5[5 * s + 1] / b0; cw[8 * s + 1] = c5[5 * s + 2] / b0;
        cw[8 * s + 2] = a1; cw[8 * s + 3] = a2;
        cw[8 * s + 4] = xfac[s]; cw[8 * s + 5] = xfac[s + 1];
    }
    f->in_gain64 = xfac[0];
    f->d_cw64 = (double *)llzs_malloc(sizeof(double) * 8 * (size_t)S);
    f->d_pd64w = (double *)llzs_malloc(sizeof(double) * 16 * (size_t)S);
    f->d_pl64w = (double *)llzs_malloc(sizeof(double) * 448 * (size_t)S);
    int rc = (f->d_cw64 && f->d_pd64w && f->d_pl64w) ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_cw64, cw, sizeof(double) * 8 * (size_t)S);
    if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_pd64w, pd, sizeof(double) * 16 * (size_t)S);
    if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_pl64w, pl, sizeof(double) * 448 * (size_t)S);
    free(t);
    return rc;
}

/* How long does the cascade remember?  The pipelined kernel may split a channel along time when there are too few
 * channels to fill the chip; a later segment then starts `warm` chunks early from the zero state.  That is sound when the
 * homogeneous response (the response to any initial state) has died out by then.  Measured here on the cascade itself:
 * unit initial outputs planted in each section in turn, zero input, the largest magnitude at any section output per
 * 1024-sample chunk; the first chunk count after which it stays below 1e-13 of its peak, or 0 if that takes more than 64
 * chunks (or the filter does not decay at all). */
static int iirm_memory_chunks(const double *c5, int S)
{
    enum { MAXC = 64, CH = 1024 };
    double env[MAXC];
    for (int k = 0; k < MAXC; k++) env[k] = 0.0;
    for (int s0 = 0; s0 < S; s0++) {
        double x1[16] = {0}, x2[16] = {0}, y1[16] = {0}, y2[16] = {0};
        y1[s0] = 1.0; y2[s0] = 1.0;
        for (int k = 0; k < MAXC; k++) {
            double m = 0.0;
            for (int i = 0; i < CH; i++) {
                double v = 0.0;
                for (int s = 0; s < S; s++) {
                    double acc = c5[5 * s] * v + c5[5 * s + 1] * x1[s] + c5[5 * s + 2] * x2[s]
                                 - c5[5 * s + 3] * y1[s] - c5[5 * s + 4] * y2[s];
                    x2[s] = x1[s]; x1[s] = v; y2[s] = y1[s]; y1[s] = acc;
                    v = acc;
                    const double a = fabs(acc);
                    if (a > m) m = a;
                }
            }
            if (!(m < 1e300)) return 0;                              /* unstable or NaN */
            if (m > env[k]) env[k] = m;
        }
    }
    double peak = 1.0;
    for (int k = 0; k < MAXC; k++) if (env[k] > peak) peak = env[k];
    int last_loud = -1;
    for (int k = 0; k < MAXC; k++) if (env[k] >= 1e-13 * peak) last_loud = k;
    if (last_loud >= MAXC - 2) return 0;                             /* still audible at the end of the probe */
    return last_loud + 2;                                            /* one chunk of margin */
}

/* May the pipelined kernel work in float32?  A rounding error e made at a section's output is filtered by the section's
 * own 1/A(z) before it reaches the next section, so float32 arithmetic adds noise of relative size eps32 * sqrt(sum g^2)
 * per section, g = impulse response of 1/A(z).  With eps32 = 6e-8 and at most 16 sections, sum g^2 <= 16 keeps the total
 * near 1e-6, a tenth of the 1e-5 tolerance (0.44-radius poles: 1.2; 0.9-radius: 30; 0.99-radius: 290 -> double).
 * Measured on each section's actual recurrence, not estimated from pole radii; anything that does not converge within
 * 4096 samples is left in double. */
static int iirm_float32_ok(const double *c5, int S)
{
    for (int s = 0; s < S; s++) {
        const double a1 = c5[5 * s + 3], a2 = c5[5 * s + 4];
        double y1 = 0.0, y2 = 0.0, energy = 0.0, tail = 0.0;
        for (int n = 0; n < 4096; n++) {
            const double y = (n == 0 ? 1.0 : 0.0) - a1 * y1 - a2 * y2;
            y2 = y1; y1 = y;
            energy += y * y;
            if (n >= 4096 - 64) tail += y * y;
        }
        if (!(energy <= 16.0) || !(tail <= 1e-20 * energy)) return 0;
        /* the feed-forward gain must not hide a cancellation either: |b| terms of ordinary size */
        if (!(fabs(c5[5 * s]) + fabs(c5[5 * s + 1]) + fabs(c5[5 * s + 2]) <= 64.0)) return 0;
    }
    return 1;
}

unsigned long llz_iir_cascade_mc_init(int channels, int stages, const double *coef)
{
    if (channels < 1 || stages < 1 || stages > 16 || !coef) {
        llzs_set_error("llz_iir_cascade_mc_init: channels %d stages %d (1..16)", channels, stages);
        return LLZ_BAD_HANDLE;
    }
    iirm_t *f = (iirm_t *)calloc(1, sizeof(*f));
    if (!f) return LLZ_BAD_HANDLE;
    f->tag = LLZ_TAG_IIRM;
    f->device = llzs_device_get();
    f->channels = channels; f->stages = stages;
    double *c5 = (double *)malloc(sizeof(double) * 5 * (size_t)stages);
    const size_t st_bytes = sizeof(double) * 4 * (size_t)stages * (size_t)channels;
    int rc = c5 ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK) {
        for (int s = 0; s < stages; s++) {                 /* {b0,b1,b2,a0,a1,a2} -> {b0,b1,b2,a1,a2}; a0 taken as 1 */
            c5[5 * s + 0] = coef[6 * s + 0]; c5[5 * s + 1] = coef[6 * s + 1]; c5[5 * s + 2] = coef[6 * s + 2];
            c5[5 * s + 3] = coef[6 * s + 4]; c5[5 * s + 4] = coef[6 * s + 5];
        }
        f->d_coef = (double *)llzs_malloc(sizeof(double) * 5 * (size_t)stages);
        f->d_state = (double *)llzs_malloc(st_bytes);
        f->d_state_alt = (double *)llzs_malloc(st_bytes);
        if (!f->d_coef || !f->d_state || !f->d_state_alt) rc = LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_coef, c5, sizeof(double) * 5 * (size_t)stages);
    if (rc == LLZ_OK) rc = llzs_memset(f->d_state, 0, st_bytes, NULL);
    if (rc == LLZ_OK) f->float32_ok = iirm_float32_ok(c5, stages) && llzs_tune(LLZS_TUNE_IIR_F64) != 1;
    if (rc == LLZ_OK) rc = iirm_build_powers(f, c5, 16);
    if (rc == LLZ_OK && llzs_tune(LLZS_TUNE_IIR_UNPACKED) < 1) rc = iirm_build_run32(f, c5);
    if (rc == LLZ_OK && llzs_tune(LLZS_TUNE_IIR_UNPACKED) < 1) rc = iirm_build_run32d(f, c5);
    if (rc == LLZ_OK) f->warm_chunks = iirm_memory_chunks(c5, stages);
    if (rc == LLZ_OK) rc = llzs_sync(NULL);
    free(c5);
    if (rc != LLZ_OK) {
        iirm_destroy(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

int llz_iir_cascade_mc_precision(unsigned long handle)
{
    if (!LLZ_HANDLE_OK(handle, iirm_t, LLZ_TAG_IIRM)) return LLZ_ERR_ARG;
    return ((iirm_t *)handle)->float32_ok ? 32 : 64;
}

void llz_iir_cascade_mc_uninit(unsigned long handle)
{
    if (LLZ_HANDLE_OK(handle, iirm_t, LLZ_TAG_IIRM)) {
        const int prev = llzs_device_enter(((iirm_t *)handle)->device);
        llzs_sync(((iirm_t *)handle)->stream);
        iirm_destroy((iirm_t *)handle);
        llzs_device_leave(prev);
    }
}

int llz_iir_cascade_mc_set_stream(unsigned long handle, void *stream)
{
    if (!LLZ_HANDLE_OK(handle, iirm_t, LLZ_TAG_IIRM)) return LLZ_ERR_ARG;
    ((iirm_t *)handle)->stream = stream;
    return LLZ_OK;
}

static int iirm_process(iirm_t *f, const float *x, float *y, int frame_len);

int llz_iir_cascade_mc(unsigned long handle, const float *x, float *y, int frame_len)
{
    if (!LLZ_HANDLE_OK(handle, iirm_t, LLZ_TAG_IIRM) || !x || !y || frame_len < 1) {
        llzs_set_error("llz_iir_cascade_mc: bad handle, buffer or frame_len");
        return LLZ_ERR_ARG;
    }
    iirm_t *f = (iirm_t *)handle;
    const int prev = llzs_device_enter(f->device);
    const int rc = iirm_process(f, x, y, frame_len);
    llzs_device_leave(prev);
    return rc;
}

static int iirm_process(iirm_t *f, const float *x, float *y, int frame_len)
{
    const size_t bytes = sizeof(float) * (size_t)f->channels * (size_t)frame_len;
    const int in_dev = llzs_is_device_ptr(x), out_dev = llzs_is_device_ptr(y);
    if (in_dev < 0 || out_dev < 0) return LLZ_ERR_ARG;            /* a buffer of another GPU: refused, message set */
    const float *d_in = x;
    float *d_out = y;
    int rc = LLZ_OK;
    if (!in_dev) {
        d_in = (const float *)llz_stage_reserve(&f->st_in, bytes);
        if (!d_in) return LLZ_ERR_NOMEM;
        rc = llzs_h2d((void *)d_in, x, bytes, f->stream);
    }
    if (rc == LLZ_OK && !out_dev) {
        d_out = (float *)llz_stage_reserve(&f->st_out, bytes);
        if (!d_out) return LLZ_ERR_NOMEM;
    }
    /* whole 1024-sample chunks go through the pipelined kernel (needs 16-byte aligned rows), the ragged remainder
     * through the one-lane-per-channel kernel; both read and write the same per-section state */
    const int aligned = (frame_len % 4 == 0) && (((size_t)d_in | (size_t)d_out) % 16 == 0);
    const int chunk = LLZS_IIR_PIPE_CHUNK;
    const int n_fast = aligned ? frame_len - frame_len % chunk : 0;
    /* short-memory cascades of up to 8 sections: a wave per (channel, time segment), all sections in registers (float32,
     * packed: 3.65 -> 2.3 ms on config 4; double: 4.76 -> 3.75 ms on the 0.99-radius set) */
    /* (needs enough (channel, segment) items to fill most of the chip, segments at least 8 x the warm-up long; measured
     * crossover against the stage pipeline, tools/iir_xover.sh: 1024 items pipeline, 2048 items wave form) */
    const long seg_items = f->warm_chunks > 0 ? (long)f->channels * (n_fast / LLZS_IIR_PIPE_CHUNK / (8 * f->warm_chunks)) : 0;
    const int min_items = llzs_tune(LLZS_TUNE_IIR_WAVE_MIN_ITEMS) >= 0 ? llzs_tune(LLZS_TUNE_IIR_WAVE_MIN_ITEMS) : 2048;
    const int wave_form = f->stages <= 8 && seg_items >= min_items && llzs_tune(LLZS_TUNE_IIR_PIPE) != 1 &&
                          (!f->float32_ok || f->d_pl32);
    /* packed float32 form (or, for cascades that need it, the double form) with 32 samples per lane on the whole
     * 2048-sample chunks, the 16-sample forms on what is left of
     * the 1024-sample chunks; every launch reads d_state and writes d_state_alt, which then swap */
    int done = 0;
    if (rc == LLZ_OK && wave_form && f->float32_ok && f->d_ph32w && n_fast >= 2048) {
        const int n32 = n_fast - n_fast % 2048;
        rc = llzs_iir_cascade_wave32_f32(d_in, d_out, f->d_pd32w, f->d_pl32w, f->d_ph32w, f->d_state, f->d_state_alt,
                                         f->channels, n32, frame_len, frame_len, f->stages, f->warm_chunks, f->in_gain32,
                                         f->stream);
        if (rc == LLZ_OK) {
            double *t = f->d_state; f->d_state = f->d_state_alt; f->d_state_alt = t;
            done = n32;
        }
    }
    if (rc == LLZ_OK && wave_form && !f->float32_ok && f->d_cw64 && n_fast >= 2048) {
        const int n32 = n_fast - n_fast % 2048;
        rc = llzs_iir_cascade_wave32_f64(d_in, d_out, f->d_cw64, f->d_pd64w, f->d_pl64w, f->d_state, f->d_state_alt,
                                         f->channels, n32, frame_len, frame_len, f->stages, f->warm_chunks, f->in_gain64,
                                         f->stream);
        if (rc == LLZ_OK) {
            double *t = f->d_state; f->d_state = f->d_state_alt; f->d_state_alt = t;
            done = n32;
        }
    }
    const int n16 = n_fast - done;
    if (rc == LLZ_OK && n16 > 0) {
        if (wave_form && f->float32_ok)
            rc = llzs_iir_cascade_wave_f32(d_in + done, d_out + done, f->d_coef32, f->d_pd32, f->d_pl32, f->d_ph32, f->d_state,
                                           f->d_state_alt, f->channels, n16, frame_len, frame_len, f->stages, f->warm_chunks,
                                           f->stream);
        else if (wave_form)
            rc = llzs_iir_cascade_wave_f64(d_in + done, d_out + done, f->d_coef, f->d_pd, f->d_pl, f->d_state, f->d_state_alt,
                                           f->channels, n16, frame_len, frame_len, f->stages, f->warm_chunks, f->stream);
        else
            rc = llzs_iir_cascade_pipe_f32(d_in + done, d_out + done, f->d_coef, f->d_pd, f->d_pl, f->d_state, f->d_state_alt,
                                           f->channels, n16, frame_len, frame_len, f->stages, f->warm_chunks, f->float32_ok,
                                           f->stream);
        if (rc == LLZ_OK) {
            double *t = f->d_state; f->d_state = f->d_state_alt; f->d_state_alt = t;
        }
    }
    if (rc == LLZ_OK && n_fast < frame_len)
        rc = llzs_iir_cascade_f32(d_in + n_fast, d_out + n_fast, f->d_coef, f->d_state, f->channels,
                                  frame_len - n_fast, frame_len, frame_len, f->stages, f->stream);
    if (rc == LLZ_OK && !out_dev) rc = llzs_d2h(y, d_out, bytes, f->stream);
    return rc == LLZ_OK ? frame_len : rc;
}


/* =====================================================================================================
 * Part 3: multi-channel GENERAL direct form I (any orders M, N up to llzs_iir_df1_mc_max_order()): the batch form of
 * llz_iir_filter itself (reference llz_iir.c:103-156), float32 in / out, double arithmetic in the reference's order.
 * ===================================================================================================== */
#define LLZ_TAG_IIRG 0x4c5a4947

typedef struct {
    int tag, device, channels, M, N, ord;
    int warm;                       /* samples after which the filter has forgotten its state to 1e-13 (0: never split time) */
    double *d_ab;                   /* a[0..ord], b[0..ord], zero padded */
    double *d_state[2];             /* [channels][2][ord + 1] delay lines, ping-pong */
    int cur;
    float *d_zero;                  /* N zeros per channel for the flush */
    llz_stage_t st_in, st_out;
    void *stream;
} iirg_t;

static void iirg_destroy(iirg_t *f)
{
    if (!f) return;
    llzs_free(f->d_ab); llzs_free(f->d_state[0]); llzs_free(f->d_state[1]); llzs_free(f->d_zero);
    llz_stage_release(&f->st_in); llz_stage_release(&f->st_out);
    f->tag = 0;
    free(f);
}

/* how long the recurrence remembers: run 1/A(z) on the host from the worst unit state (every y delay = 1) with zero input
 * and find the last sample whose magnitude exceeds 1e-13 of the largest seen; 0 when it has not died out within `limit` */
static int iirg_probe_memory(int M, const double *a, int limit)
{
    if (M == 0) return 1;
    double y[64];
    for (int k = 0; k < M; k++) y[k] = 1.0;
    double peak = 1.0;
    int last = 0;
    for (int t = 0; t < limit; t++) {
        double acc = 0.0;
        for (int k = 1; k <= M; k++) acc -= a[k] * y[k - 1];
        for (int k = M - 1; k >= 1; k--) y[k] = y[k - 1];
        y[0] = acc;
        const double m = fabs(acc);
        if (!(m < 1e300)) return 0;                                 /* unstable */
        if (m > peak) peak = m;
        if (m > 1e-13 * peak) last = t;
    }
    return last < limit - limit / 8 ? last + 1 : 0;
}

unsigned long llz_iir_mc_init(int channels, int M, const double *a, int N, const double *b)
{
    const int ord = llzs_iir_df1_mc_max_order();
    if (channels < 1 || M < 0 || N < 0 || M > ord || N > ord || !a) {
        llzs_set_error("llz_iir_mc_init: channels %d M %d N %d (orders 0..%d) or NULL a", channels, M, N, ord);
        return LLZ_BAD_HANDLE;
    }
    iirg_t *f = (iirg_t *)calloc(1, sizeof(*f));
    double *ab = (double *)calloc(2 * ((size_t)ord + 1), sizeof(double));
    int rc = (f && ab) ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK) {
        f->tag = LLZ_TAG_IIRG; f->device = llzs_device_get(); f->channels = channels; f->M = M; f->N = N; f->ord = ord;
        for (int k = 0; k <= M; k++) ab[k] = a[k];
        for (int k = 0; b && k <= N; k++) ab[ord + 1 + k] = b[k];                  /* b == NULL -> zeros (llz_iir.c:54-59) */
        const int mem = iirg_probe_memory(M, a, 1 << 16);
        f->warm = mem ? mem + N : 0;
        const size_t sbytes = sizeof(double) * (size_t)channels * 2 * ((size_t)ord + 1);
        f->d_ab = (double *)llzs_malloc(sizeof(double) * 2 * ((size_t)ord + 1));
        f->d_state[0] = (double *)llzs_malloc(sbytes);
        f->d_state[1] = (double *)llzs_malloc(sbytes);
        f->d_zero = (float *)llzs_malloc(sizeof(float) * (size_t)channels * (size_t)(N > 0 ? N : 1));
        if (!f->d_ab || !f->d_state[0] || !f->d_state[1] || !f->d_zero || f->device < 0) rc = LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_ab, ab, sizeof(double) * 2 * ((size_t)ord + 1));
        if (rc == LLZ_OK) rc = llzs_memset(f->d_state[0], 0, sbytes, NULL);
        if (rc == LLZ_OK) rc = llzs_memset(f->d_state[1], 0, sbytes, NULL);
        if (rc == LLZ_OK) rc = llzs_memset(f->d_zero, 0, sizeof(float) * (size_t)channels * (size_t)(N > 0 ? N : 1), NULL);
        if (rc == LLZ_OK) rc = llzs_sync(NULL);
    }
    free(ab);
    if (rc != LLZ_OK) {
        if (f && f->tag) iirg_destroy(f); else free(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

void llz_iir_mc_uninit(unsigned long handle)
{
    if (!LLZ_HANDLE_OK(handle, iirg_t, LLZ_TAG_IIRG)) return;
    iirg_t *f = (iirg_t *)handle;
    const int prev = llzs_device_enter(f->device);
    llzs_sync(f->stream);
    iirg_destroy(f);
    llzs_device_leave(prev);
}

int llz_iir_mc_set_stream(unsigned long handle, void *stream)
{
    if (!LLZ_HANDLE_OK(handle, iirg_t, LLZ_TAG_IIRG)) return LLZ_ERR_ARG;
    ((iirg_t *)handle)->stream = stream;
    return LLZ_OK;
}

static int iirg_launch(iirg_t *f, const float *d_in, float *d_out, int n)
{
    /* time segments: enough (channel, segment) lanes to fill the chip (~64 K), each at least 8 x the filter's memory long */
    int segs = 1;
    if (f->warm > 0) {
        const int tune = llzs_tune(LLZS_TUNE_IIR_SEGS);
        const long want = tune > 0 ? tune : (65536 + f->channels - 1) / f->channels;
        const long most = (long)n / (8L * f->warm);
        segs = (int)(want < most ? want : most);
        if (segs < 1) segs = 1;
    }
    const int rc = llzs_iir_df1_mc_f32(d_in, d_out, f->d_ab, f->d_state[f->cur], f->d_state[f->cur ^ 1], f->channels, n, n, n,
                                       f->M, f->N, segs, f->warm, f->stream);
    if (rc == LLZ_OK) f->cur ^= 1;
    return rc;
}

int llz_iir_mc(unsigned long handle, const float *x, float *y, int frame_len)
{
    if (!LLZ_HANDLE_OK(handle, iirg_t, LLZ_TAG_IIRG) || !x || !y || frame_len < 1 || x == y) {
        llzs_set_error("llz_iir_mc: bad handle, NULL buffer, in-place call or frame_len %d", frame_len);
        return LLZ_ERR_ARG;
    }
    iirg_t *f = (iirg_t *)handle;
    const int prev = llzs_device_enter(f->device);
    const size_t bytes = sizeof(float) * (size_t)f->channels * (size_t)frame_len;
    const int in_dev = llzs_is_device_ptr(x), out_dev = llzs_is_device_ptr(y);
    const float *d_in = x;
    float *d_out = y;
    int rc = (in_dev < 0 || out_dev < 0) ? LLZ_ERR_ARG : LLZ_OK;
    if (rc == LLZ_OK && !in_dev) {
        d_in = (const float *)llz_stage_reserve(&f->st_in, bytes);
        rc = d_in ? llzs_h2d((void *)d_in, x, bytes, f->stream) : LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK && !out_dev) {
        d_out = (float *)llz_stage_reserve(&f->st_out, bytes);
        if (!d_out) rc = LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK) rc = iirg_launch(f, d_in, d_out, frame_len);
    if (rc == LLZ_OK && !out_dev) rc = llzs_d2h(y, d_out, bytes, f->stream);
    llzs_device_leave(prev);
    return rc == LLZ_OK ? frame_len : rc;
}

/* N more samples of x = 0 per channel (llz_iir.c:147-156); returns N */
int llz_iir_mc_flush(unsigned long handle, float *y)
{
    if (!LLZ_HANDLE_OK(handle, iirg_t, LLZ_TAG_IIRG) || !y) {
        llzs_set_error("llz_iir_mc_flush: bad handle or buffer");
        return LLZ_ERR_ARG;
    }
    iirg_t *f = (iirg_t *)handle;
    if (f->N == 0) return 0;
    const int prev = llzs_device_enter(f->device);
    const size_t bytes = sizeof(float) * (size_t)f->channels * (size_t)f->N;
    const int out_dev = llzs_is_device_ptr(y);
    float *d_out = y;
    int rc = out_dev < 0 ? LLZ_ERR_ARG : LLZ_OK;
    if (rc == LLZ_OK && !out_dev) {
        d_out = (float *)llz_stage_reserve(&f->st_out, bytes);
        if (!d_out) rc = LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK) rc = iirg_launch(f, f->d_zero, d_out, f->N);
    if (rc == LLZ_OK && !out_dev) rc = llzs_d2h(y, d_out, bytes, f->stream);
    llzs_device_leave(prev);
    return rc == LLZ_OK ? f->N : rc;
}
