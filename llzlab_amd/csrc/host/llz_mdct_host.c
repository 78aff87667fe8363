/*
 * llz_mdct_host.c -- handle layer of the MDCT: the reference's symbols
 * (reference libllzfilter/llz_mdct.c:97-620) and the float32 batch extension.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../../include/llz_mdct.h"
#include "../../../include/llz_fft.h"
#include "llz_host.h"

#define LLZ_TAG_MDCT 0x4c5a4d31
#define LLZ_TAG_MDCB 0x4c5a4d42

/* ---- windows (host, llz_mdct.c:97-182) ---- */

int llz_mdct_sine(double *w, int N)
{
    for (int n = 0; n < N; n++) {
        const double tmp = (M_PI / N) * (n + 0.5);
        w[n] = sin(tmp);
    }
    return N;
}

static double mdct_bessel(double x)
{
    double xh = (double)0.5 * x, sum = 1.0, pw = 1.0, ds = 1.0;
    int k = 0;
    while (ds > sum * 1E-16) {
        ++k;
        pw = pw * (xh / k);
        ds = pw * pw;
        sum = sum + ds;
    }
    return sum;
}

int llz_mdct_kbd(double *w, int N, double alpha)
{
    const int N2 = N >> 1, K = N2 + 1;
    double *w1 = (double *)malloc(sizeof(double) * (size_t)K);
    if (!w1) return -1;
    const double beta = alpha * M_PI;
    for (int i = 0; i < K; i++) {                                   /* kaiser_beta(w1, N2+1, alpha*pi) */
        const double Ib = mdct_bessel(beta);
        const double x = (double)((2. * i / (K - 1)) - 1);
        const double Ia = mdct_bessel(beta * (double)sqrt(1. - x * x));
        w1[i] = (double)(Ia / Ib);
    }
    double sum = 0.0, tmp = 0.0;
    for (int i = 0; i < K; i++) sum += w1[i];
    sum = 1.0 / sum;
    for (int i = 0, j = N - 1; i < N2; i++, j--) {
        tmp += w1[i];
        w[i] = w[j] = sqrt(tmp * sum);
    }
    free(w1);
    return N;
}

/* ---- Part 1: reference symbols ---- */

typedef struct {
    int tag, type, length;
    unsigned long h_fft;
    double *fft_buf;
    /* type 0: cosine matrices on the device, vectors staged through d_x / d_y */
    double *d_cos_pos, *d_cos_inv, *d_x, *d_y;
    /* type 1 */
    double *pre_c_pos, *pre_s_pos, *c_pos, *s_pos, *pre_c_inv, *pre_s_inv, *c_inv, *s_inv;
    /* type 2 */
    double *tw_c, *tw_s, *rot, sqrt_cof;
} mdct1_t;

static void mdct1_destroy(mdct1_t *f)
{
    if (!f) return;
    if (f->h_fft && f->h_fft != LLZ_BAD_HANDLE) llz_fft_uninit(f->h_fft);
    free(f->fft_buf);
    llzs_free(f->d_cos_pos); llzs_free(f->d_cos_inv); llzs_free(f->d_x); llzs_free(f->d_y);
    free(f->pre_c_pos); free(f->pre_s_pos); free(f->c_pos); free(f->s_pos);
    free(f->pre_c_inv); free(f->pre_s_inv); free(f->c_inv); free(f->s_inv);
    free(f->tw_c); free(f->tw_s); free(f->rot);
    f->tag = 0;
    free(f);
}

unsigned long llz_mdct_init(int type, int size)
{
    if (size < 4 || (type != MDCT_ORIGIN && type != MDCT_FFT && type != MDCT_FFT4)) {
        llzs_set_error("llz_mdct_init: type %d len %d", type, size);
        return LLZ_BAD_HANDLE;
    }
    int base = (int)(log(size) / log(2));                           /* llz_mdct.c:375-379 */
    if ((1 << base) < size) base += 1;
    const int length = 1 << base;
    const int limit = type == MDCT_ORIGIN ? 2048 : (type == MDCT_FFT ? 4096 : 16384);
    if (length > limit || (type == MDCT_FFT4 && length < 8)) {
        llzs_set_error("llz_mdct_init: length %d out of range for type %d (at most %d)", length, type, limit);
        return LLZ_BAD_HANDLE;
    }
    mdct1_t *f = (mdct1_t *)calloc(1, sizeof(*f));
    if (!f) return LLZ_BAD_HANDLE;
    f->tag = LLZ_TAG_MDCT; f->type = type; f->length = length;
    int rc = LLZ_OK;
    if (type == MDCT_ORIGIN) {                                      /* llz_mdct.c:384-403 */
        const size_t cnt = (size_t)(length >> 1) * length;
        double *pos = (double *)malloc(sizeof(double) * cnt), *inv = (double *)malloc(sizeof(double) * cnt);
        f->d_cos_pos = (double *)llzs_malloc(sizeof(double) * cnt);
        f->d_cos_inv = (double *)llzs_malloc(sizeof(double) * cnt);
        f->d_x = (double *)llzs_malloc(sizeof(double) * (size_t)length);
        f->d_y = (double *)llzs_malloc(sizeof(double) * (size_t)length);
        if (!pos || !inv || !f->d_cos_pos || !f->d_cos_inv || !f->d_x || !f->d_y) rc = LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) {
            for (int k = 0; k < (length >> 1); k++)
                for (int n = 0; n < length; n++) {
                    const double tmp = (M_PI / (2 * length)) * (2 * n + 1 + (length >> 1)) * (2 * k + 1);
                    pos[(size_t)k * length + n] = inv[(size_t)n * (length >> 1) + k] = cos(tmp);
                }
            rc = llzs_h2d(f->d_cos_pos, pos, sizeof(double) * cnt, NULL);
            if (rc == LLZ_OK) rc = llzs_h2d(f->d_cos_inv, inv, sizeof(double) * cnt, NULL);
        }
        free(pos); free(inv);
    } else if (type == MDCT_FFT) {                                  /* llz_mdct.c:404-448 */
        const double n0 = ((double)length / 2 + 1) / 2;
        f->h_fft = llz_fft_init(length);
        f->fft_buf = (double *)malloc(sizeof(double) * (size_t)length * 2);
        f->pre_c_pos = (double *)malloc(sizeof(double) * (size_t)length);
        f->pre_s_pos = (double *)malloc(sizeof(double) * (size_t)length);
        f->c_pos = (double *)malloc(sizeof(double) * (size_t)(length >> 1));
        f->s_pos = (double *)malloc(sizeof(double) * (size_t)(length >> 1));
        f->pre_c_inv = (double *)malloc(sizeof(double) * (size_t)length);
        f->pre_s_inv = (double *)malloc(sizeof(double) * (size_t)length);
        f->c_inv = (double *)malloc(sizeof(double) * (size_t)length);
        f->s_inv = (double *)malloc(sizeof(double) * (size_t)length);
        if (f->h_fft == LLZ_BAD_HANDLE || !f->fft_buf || !f->pre_c_pos || !f->pre_s_pos || !f->c_pos || !f->s_pos ||
            !f->pre_c_inv || !f->pre_s_inv || !f->c_inv || !f->s_inv) rc = LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) {
            for (int k = 0; k < length; k++) {
                f->pre_c_pos[k] = cos(-(M_PI * k) / length);
                f->pre_s_pos[k] = sin(-(M_PI * k) / length);
            }
            for (int k = 0; k < (length >> 1); k++) {
                f->c_pos[k] = cos(-2 * M_PI * n0 * (k + 0.5) / length);
                f->s_pos[k] = sin(-2 * M_PI * n0 * (k + 0.5) / length);
            }
            for (int k = 0; k < length; k++) {
                f->pre_c_inv[k] = cos((2 * M_PI * k * n0) / length);
                f->pre_s_inv[k] = sin((2 * M_PI * k * n0) / length);
            }
            for (int k = 0; k < length; k++) {
                f->c_inv[k] = cos(M_PI * (k + n0) / length);
                f->s_inv[k] = sin(M_PI * (k + n0) / length);
            }
        }
    } else {                                                        /* llz_mdct.c:449-467 */
        f->h_fft = llz_fft_init(length >> 2);
        f->fft_buf = (double *)malloc(sizeof(double) * (size_t)(length >> 1));
        f->sqrt_cof = 1. / sqrt(length);
        f->rot = (double *)calloc((size_t)length, sizeof(double));
        f->tw_c = (double *)malloc(sizeof(double) * (size_t)(length >> 2));
        f->tw_s = (double *)malloc(sizeof(double) * (size_t)(length >> 2));
        if (f->h_fft == LLZ_BAD_HANDLE || !f->fft_buf || !f->rot || !f->tw_c || !f->tw_s) rc = LLZ_ERR_NOMEM;
        if (rc == LLZ_OK)
            for (int k = 0; k < (length >> 2); k++) {
                f->tw_c[k] = cos(-2 * M_PI * (k + 0.125) / length);
                f->tw_s[k] = sin(-2 * M_PI * (k + 0.125) / length);
            }
    }
    if (rc != LLZ_OK) {
        mdct1_destroy(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

void llz_mdct_uninit(unsigned long handle)
{
    if (LLZ_HANDLE_OK(handle, mdct1_t, LLZ_TAG_MDCT)) mdct1_destroy((mdct1_t *)handle);
}

/* the defining sums on the device: y[r] = sum_c x[c] * A[r][c] */
static int mdct0_sums(mdct1_t *f, const double *d_A, const double *x, double *y, int rows, int cols)
{
    int rc = llzs_h2d(f->d_x, x, sizeof(double) * (size_t)cols, NULL);
    if (rc == LLZ_OK) rc = llzs_matvec_exact_f64(d_A, f->d_x, f->d_y, rows, cols, NULL);
    if (rc == LLZ_OK) rc = llzs_d2h(y, f->d_y, sizeof(double) * (size_t)rows, NULL);
    return rc;
}

void llz_mdct(unsigned long handle, double *x, double *X)
{
    if (!LLZ_HANDLE_OK(handle, mdct1_t, LLZ_TAG_MDCT) || !x || !X) {
        llzs_set_error("llz_mdct: bad handle or arguments");
        return;                                                     /* void in the reference ABI */
    }
    mdct1_t *f = (mdct1_t *)handle;
    const int N = f->length, N2 = N >> 1, N4 = N >> 2;
    if (f->type == MDCT_ORIGIN) {                                   /* llz_mdct.c:185-202 */
        (void)mdct0_sums(f, f->d_cos_pos, x, X, N2, N);
    } else if (f->type == MDCT_FFT) {                               /* llz_mdct.c:225-241 */
        for (int k = 0; k < N; k++) {
            f->fft_buf[k + k] = x[k] * f->pre_c_pos[k];
            f->fft_buf[k + k + 1] = x[k] * f->pre_s_pos[k];
        }
        llz_fft(f->h_fft, f->fft_buf);
        for (int k = 0; k < N2; k++)
            X[k] = f->fft_buf[k + k] * f->c_pos[k] - f->fft_buf[k + k + 1] * f->s_pos[k];
    } else {                                                        /* llz_mdct.c:266-303 */
        double *rot = f->rot;
        memset(rot, 0, sizeof(double) * (size_t)f->length);
        for (int k = 0; k < N4; k++) rot[k] = -x[k + 3 * N4];
        for (int k = N4; k < N; k++) rot[k] = x[k - N4];
        for (int k = 0; k < N4; k++) {
            const double re = rot[2 * k] - rot[N - 1 - 2 * k];
            const double im = rot[N2 - 1 - 2 * k] - rot[N2 + 2 * k];
            f->fft_buf[k + k] = 0.5 * (re * f->tw_c[k] - im * f->tw_s[k]);
            f->fft_buf[k + k + 1] = 0.5 * (re * f->tw_s[k] + im * f->tw_c[k]);
        }
        llz_fft(f->h_fft, f->fft_buf);
        for (int k = 0; k < N4; k++) {
            const double re = f->fft_buf[k + k], im = f->fft_buf[k + k + 1];
            X[2 * k] = 2 * (re * f->tw_c[k] - im * f->tw_s[k]);
            X[N2 - 1 - 2 * k] = -2 * (re * f->tw_s[k] + im * f->tw_c[k]);
        }
    }
}

void llz_imdct(unsigned long handle, double *X, double *x)
{
    if (!LLZ_HANDLE_OK(handle, mdct1_t, LLZ_TAG_MDCT) || !x || !X) {
        llzs_set_error("llz_imdct: bad handle or arguments");
        return;
    }
    mdct1_t *f = (mdct1_t *)handle;
    const int N = f->length, N2 = N >> 1, N4 = N >> 2;
    if (f->type == MDCT_ORIGIN) {                                   /* llz_mdct.c:204-222 */
        if (mdct0_sums(f, f->d_cos_inv, X, x, N, N2) == LLZ_OK)
            for (int n = 0; n < N; n++) x[n] = (x[n] * 4) / N;
    } else if (f->type == MDCT_FFT) {                               /* llz_mdct.c:243-264 */
        for (int k = 0; k < N2; k++) {
            f->fft_buf[k + k] = X[k] * f->pre_c_inv[k];
            f->fft_buf[k + k + 1] = X[k] * f->pre_s_inv[k];
        }
        for (int k = N2, i = N2 - 1; k < N; k++, i--) {
            f->fft_buf[k + k] = -X[i] * f->pre_c_inv[k];
            f->fft_buf[k + k + 1] = -X[i] * f->pre_s_inv[k];
        }
        llz_ifft(f->h_fft, f->fft_buf);
        for (int k = 0; k < N; k++)
            x[k] = 2 * (f->fft_buf[k + k] * f->c_inv[k] - f->fft_buf[k + k + 1] * f->s_inv[k]);
    } else {                                                        /* llz_mdct.c:305-353 (a forward llz_fft here too) */
        double *rot = f->rot;
        const double cof = f->sqrt_cof;
        memset(rot, 0, sizeof(double) * (size_t)f->length);
        for (int k = 0; k < N4; k++) {
            const double re = X[2 * k], im = X[N2 - 1 - 2 * k];
            f->fft_buf[k + k] = 0.5 * (re * f->tw_c[k] - im * f->tw_s[k]);
            f->fft_buf[k + k + 1] = 0.5 * (re * f->tw_s[k] + im * f->tw_c[k]);
        }
        llz_fft(f->h_fft, f->fft_buf);
        for (int k = 0; k < N4; k++) {
            const double re = f->fft_buf[k + k], im = f->fft_buf[k + k + 1];
            f->fft_buf[k + k] = 8 * cof * (re * f->tw_c[k] - im * f->tw_s[k]);
            f->fft_buf[k + k + 1] = 8 * cof * (re * f->tw_s[k] + im * f->tw_c[k]);
        }
        for (int k = 0; k < N4; k++) {
            rot[2 * k] = f->fft_buf[k + k];
            rot[N2 + 2 * k] = f->fft_buf[k + k + 1];
        }
        for (int k = 1; k < N; k += 2) rot[k] = -rot[N - 1 - k];
        for (int k = 0; k < 3 * N4; k++) x[k] = rot[N4 + k] * cof;
        for (int k = 3 * N4; k < N; k++) x[k] = -rot[k - 3 * N4] * cof;
    }
}

/* ---- Part 2: batch extension ---- */

typedef struct {
    int tag, length;
    float *d_tc, *d_ts, *d_cs;
    llz_stage_t st_in, st_out;
    void *stream;
} mdcb_t;

static void mdcb_destroy(mdcb_t *f)
{
    if (!f) return;
    llzs_free(f->d_tc); llzs_free(f->d_ts); llzs_free(f->d_cs);
    llz_stage_release(&f->st_in); llz_stage_release(&f->st_out);
    f->tag = 0;
    free(f);
}

unsigned long llz_mdct_batch_init(int len)
{
    if (len < 32 || len > 8192 || (len & (len - 1))) {
        llzs_set_error("llz_mdct_batch_init: len %d must be a power of two in 32..8192", len);
        return LLZ_BAD_HANDLE;
    }
    const int N4 = len >> 2;
    mdcb_t *f = (mdcb_t *)calloc(1, sizeof(*f));
    float *tab = (float *)malloc(sizeof(float) * 4 * (size_t)N4);
    int rc = (f && tab) ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK) {
        f->tag = LLZ_TAG_MDCB; f->length = len;
        for (int k = 0; k < N4; k++) {
            tab[k] = (float)cos(-2 * M_PI * (k + 0.125) / len);           /* llz_mdct.c:459-462 */
            tab[N4 + k] = (float)sin(-2 * M_PI * (k + 0.125) / len);
            const double ang = (double)(2 * M_PI * k) / N4;               /* llz_fft.c:222-229 for size N/4 */
            tab[2 * N4 + k] = (float)cos(ang);
            tab[3 * N4 + k] = (float)sin(ang);
        }
        f->d_tc = (float *)llzs_malloc(sizeof(float) * (size_t)N4);
        f->d_ts = (float *)llzs_malloc(sizeof(float) * (size_t)N4);
        f->d_cs = (float *)llzs_malloc(sizeof(float) * 2 * (size_t)N4);
        if (!f->d_tc || !f->d_ts || !f->d_cs) rc = LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) rc = llzs_h2d(f->d_tc, tab, sizeof(float) * (size_t)N4, NULL);
        if (rc == LLZ_OK) rc = llzs_h2d(f->d_ts, tab + N4, sizeof(float) * (size_t)N4, NULL);
        if (rc == LLZ_OK) rc = llzs_h2d(f->d_cs, tab + 2 * N4, sizeof(float) * 2 * (size_t)N4, NULL);
    }
    free(tab);
    if (rc != LLZ_OK) {
        if (f && f->tag) mdcb_destroy(f); else free(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

void llz_mdct_batch_uninit(unsigned long handle)
{
    if (LLZ_HANDLE_OK(handle, mdcb_t, LLZ_TAG_MDCB)) {
        llzs_sync(((mdcb_t *)handle)->stream);
        mdcb_destroy((mdcb_t *)handle);
    }
}

int llz_mdct_batch_set_stream(unsigned long handle, void *stream)
{
    if (!LLZ_HANDLE_OK(handle, mdcb_t, LLZ_TAG_MDCB)) return LLZ_ERR_ARG;
    ((mdcb_t *)handle)->stream = stream;
    return LLZ_OK;
}

static int mdcb_run(unsigned long handle, const float *in, float *out, int count, int inverse, const char *who)
{
    if (!LLZ_HANDLE_OK(handle, mdcb_t, LLZ_TAG_MDCB) || !in || !out || count < 1) {
        llzs_set_error("%s: bad handle or arguments", who);
        return LLZ_ERR_ARG;
    }
    mdcb_t *f = (mdcb_t *)handle;
    const size_t full = sizeof(float) * (size_t)count * f->length, half = full / 2;
    const size_t ib = inverse ? half : full, ob = inverse ? full : half;
    const float *d_in = in;
    float *d_out = out;
    int rc = LLZ_OK;
    if (!llzs_is_device_ptr(in)) {
        d_in = (const float *)llz_stage_reserve(&f->st_in, ib);
        rc = d_in ? llzs_h2d((void *)d_in, in, ib, f->stream) : LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK && !llzs_is_device_ptr(out)) {
        d_out = (float *)llz_stage_reserve(&f->st_out, ob);
        if (!d_out) rc = LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK) rc = llzs_mdct4_f32(d_in, d_out, count, f->length, f->d_tc, f->d_ts, f->d_cs, inverse, f->stream);
    if (rc == LLZ_OK && d_out != out) rc = llzs_d2h(out, d_out, ob, f->stream);
    return rc == LLZ_OK ? count : rc;
}

int llz_mdct_batch(unsigned long handle, const float *x, float *X, int count)
{
    return mdcb_run(handle, x, X, count, 0, "llz_mdct_batch");
}

int llz_imdct_batch(unsigned long handle, const float *X, float *x, int count)
{
    return mdcb_run(handle, X, x, count, 1, "llz_imdct_batch");
}
