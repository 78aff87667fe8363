/*
 * llz_mdct_host.c -- handle layer of the MDCT: the reference's symbols
 * (reference libllzfilter/llz_mdct.c:97-620) and the float32 batch extension.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../../include/llz_mdct.h"
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

/* ---- Part 1: reference symbols ----
 * All three algorithms of llz_mdct.c run on the device in the reference's rounding order.  A frame travels host -> device
 * once, goes through (twiddle step, exact-order transform, twiddle step) or the defining sums, and comes back once; the
 * host only builds the angle tables (libm, the reference's angle expressions, so that the table values are the reference's
 * to the last bit) and moves the frame. */

enum { ROT_FWD_PRE = 0, ROT_FWD_POST = 1, ROT_INV_PRE = 2, ROT_INV_POST = 3 };

typedef struct {
    int tag, form, length, device;  /* form: MDCT_ORIGIN (defining sums), MDCT_FFT (N-point), MDCT_FFT4 (N/4-point) */
    int fft_size;                   /* points of the transform inside the FFT forms */
    double norm;                    /* 1 / sqrt(length): the N/4-point form's inverse scale */
    double *d_sum_fwd, *d_sum_inv;  /* MDCT_ORIGIN: [N/2][N] and [N][N/2] cosine kernels */
    double *d_rot[4];               /* FFT forms: (cos, sin) pair tables of the four twiddle steps */
    double *d_fft_cs;               /* fft_size cos then fft_size sin of 2 pi i / fft_size (llz_fft.c:222-229) */
    double *d_frame, *d_bins, *d_work;   /* N time samples, N/2 coefficients, up to N complex points */
} mdct1_t;

static void mdct1_destroy(mdct1_t *f)
{
    if (!f) return;
    llzs_free(f->d_sum_fwd); llzs_free(f->d_sum_inv);
    for (int t = 0; t < 4; t++) {
        int shared = 0;
        for (int u = 0; u < t; u++) shared |= (f->d_rot[u] == f->d_rot[t]);
        if (!shared) llzs_free(f->d_rot[t]);
    }
    llzs_free(f->d_fft_cs); llzs_free(f->d_frame); llzs_free(f->d_bins); llzs_free(f->d_work);
    f->tag = 0;
    free(f);
}

/* angle of entry k of a twiddle table; the expressions are the reference's (llz_mdct.c:418-441, :459-462), evaluated in
 * its order, because the last bit of cos/sin of a differently rounded angle would differ */
static double rot_angle(int form, int step, int k, int length)
{
    if (form == MDCT_FFT4) return -2 * M_PI * (k + 0.125) / length;
    const double n0 = ((double)length / 2 + 1) / 2;
    switch (step) {
    case ROT_FWD_PRE:  return -(M_PI * k) / length;
    case ROT_FWD_POST: return -2 * M_PI * n0 * (k + 0.5) / length;
    case ROT_INV_PRE:  return (2 * M_PI * k * n0) / length;
    default:           return M_PI * (k + n0) / length;
    }
}

static double *rot_table_upload(int form, int step, int count, int length)
{
    double *host = (double *)malloc(sizeof(double) * 2 * (size_t)count);
    double *dev = (double *)llzs_malloc(sizeof(double) * 2 * (size_t)count);
    if (host && dev) {
        for (int k = 0; k < count; k++) {
            const double ang = rot_angle(form, step, k, length);
            host[2 * k] = cos(ang);
            host[2 * k + 1] = sin(ang);
        }
        if (llzs_h2d(dev, host, sizeof(double) * 2 * (size_t)count, NULL) != LLZ_OK) { llzs_free(dev); dev = NULL; }
    } else {
        llzs_free(dev);
        dev = NULL;
    }
    free(host);
    return dev;
}

static int mdct1_build_sums(mdct1_t *f)
{
    const int N = f->length, K = N >> 1;
    const size_t cnt = (size_t)K * N;
    double *fwd = (double *)malloc(sizeof(double) * cnt), *inv = (double *)malloc(sizeof(double) * cnt);
    f->d_sum_fwd = (double *)llzs_malloc(sizeof(double) * cnt);
    f->d_sum_inv = (double *)llzs_malloc(sizeof(double) * cnt);
    int rc = (fwd && inv && f->d_sum_fwd && f->d_sum_inv) ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK) {
        for (int k = 0; k < K; k++)
            for (int n = 0; n < N; n++) {
                const double ang = (M_PI / (2 * N)) * (2 * n + 1 + K) * (2 * k + 1);      /* llz_mdct.c:392-399 */
                fwd[(size_t)k * N + n] = inv[(size_t)n * K + k] = cos(ang);
            }
        rc = llzs_h2d(f->d_sum_fwd, fwd, sizeof(double) * cnt, NULL);
        if (rc == LLZ_OK) rc = llzs_h2d(f->d_sum_inv, inv, sizeof(double) * cnt, NULL);
    }
    free(fwd); free(inv);
    return rc;
}

static int mdct1_build_fft_form(mdct1_t *f)
{
    const int N = f->length;
    f->fft_size = f->form == MDCT_FFT ? N : N >> 2;
    if (f->form == MDCT_FFT) {
        f->d_rot[ROT_FWD_PRE] = rot_table_upload(MDCT_FFT, ROT_FWD_PRE, N, N);
        f->d_rot[ROT_FWD_POST] = rot_table_upload(MDCT_FFT, ROT_FWD_POST, N >> 1, N);
        f->d_rot[ROT_INV_PRE] = rot_table_upload(MDCT_FFT, ROT_INV_PRE, N, N);
        f->d_rot[ROT_INV_POST] = rot_table_upload(MDCT_FFT, ROT_INV_POST, N, N);
    } else {
        double *t = rot_table_upload(MDCT_FFT4, 0, N >> 2, N);          /* one table serves all four steps */
        for (int s = 0; s < 4; s++) f->d_rot[s] = t;
        f->norm = 1. / sqrt(N);
    }
    for (int s = 0; s < 4; s++)
        if (!f->d_rot[s]) return LLZ_ERR_NOMEM;
    const int F = f->fft_size;
    double *cs = (double *)malloc(sizeof(double) * 2 * (size_t)F);
    f->d_fft_cs = (double *)llzs_malloc(sizeof(double) * 2 * (size_t)F);
    f->d_work = (double *)llzs_malloc(sizeof(double) * 2 * (size_t)F);
    int rc = (cs && f->d_fft_cs && f->d_work) ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK) {
        for (int i = 0; i < F; i++) {
            const double ang = (double)(2 * M_PI * i) / F;
            cs[i] = cos(ang);
            cs[F + i] = sin(ang);
        }
        rc = llzs_h2d(f->d_fft_cs, cs, sizeof(double) * 2 * (size_t)F, NULL);
    }
    free(cs);
    return rc;
}

unsigned long llz_mdct_init(int type, int size)
{
    if (size < 4 || (type != MDCT_ORIGIN && type != MDCT_FFT && type != MDCT_FFT4)) {
        llzs_set_error("llz_mdct_init: type %d len %d", type, size);
        return LLZ_BAD_HANDLE;
    }
    int log2len = (int)(log(size) / log(2));                        /* next power of two, as llz_mdct.c:375-379 */
    if ((1 << log2len) < size) log2len += 1;
    const int length = 1 << log2len;
    const int limit = type == MDCT_ORIGIN ? 2048 : (type == MDCT_FFT ? 4096 : 16384);
    if (length > limit || (type == MDCT_FFT4 && length < 8)) {
        llzs_set_error("llz_mdct_init: length %d out of range for type %d (at most %d)", length, type, limit);
        return LLZ_BAD_HANDLE;
    }
    mdct1_t *f = (mdct1_t *)calloc(1, sizeof(*f));
    if (!f) return LLZ_BAD_HANDLE;
    f->tag = LLZ_TAG_MDCT; f->form = type; f->length = length; f->device = llzs_device_get();
    f->d_frame = (double *)llzs_malloc(sizeof(double) * (size_t)length);
    f->d_bins = (double *)llzs_malloc(sizeof(double) * (size_t)length);
    int rc = (f->d_frame && f->d_bins) ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK) rc = type == MDCT_ORIGIN ? mdct1_build_sums(f) : mdct1_build_fft_form(f);
    if (rc != LLZ_OK) {
        mdct1_destroy(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

void llz_mdct_uninit(unsigned long handle)
{
    if (!LLZ_HANDLE_OK(handle, mdct1_t, LLZ_TAG_MDCT)) return;
    const int prev = llzs_device_enter(((mdct1_t *)handle)->device);
    llzs_sync(NULL);
    mdct1_destroy((mdct1_t *)handle);
    llzs_device_leave(prev);
}

/* one direction of one frame between two DEVICE buffers (the frame handles of llz_asmodel_host.c chain this with their own
 * windowing kernels; d_src is not modified, d_dst != d_src) */
static int mdct1_on_device(mdct1_t *f, const double *d_src, double *d_dst, int inverse)
{
    const int N = f->length, K = N >> 1;
    const int n_src = inverse ? K : N, n_dst = inverse ? N : K;
    int rc;
    if (f->form == MDCT_ORIGIN) {                                    /* llz_mdct.c:185-222: the defining sums */
        rc = llzs_matvec_exact_f64(inverse ? f->d_sum_inv : f->d_sum_fwd, d_src, d_dst, n_dst, n_src, NULL);
        if (rc == LLZ_OK && inverse) rc = llzs_scale_4_over_n_f64(d_dst, N, (double)N, NULL);   /* llz_mdct.c:219-220 */
        return rc;
    }
    const int quarter = f->form == MDCT_FFT4;
    /* the N-point form inverts with the inverse transform (llz_mdct.c:258); the N/4-point form uses the FORWARD
     * transform in both directions (llz_mdct.c:291, :328) */
    const int fft_inverse = inverse && !quarter;
    rc = llzs_mdct_rot_f64(quarter, 0, d_src, f->d_work, f->d_rot[inverse ? ROT_INV_PRE : ROT_FWD_PRE], N, inverse,
                           f->norm, NULL);
    if (rc == LLZ_OK) rc = llzs_fft_f64(f->d_work, f->fft_size, f->d_fft_cs, fft_inverse, NULL);
    if (rc == LLZ_OK)
        rc = llzs_mdct_rot_f64(quarter, 1, f->d_work, d_dst, f->d_rot[inverse ? ROT_INV_POST : ROT_FWD_POST], N,
                               inverse, f->norm, NULL);
    return rc;
}

int llz_host_mdct_on_device(unsigned long handle, const double *d_src, double *d_dst, int inverse)
{
    if (!LLZ_HANDLE_OK(handle, mdct1_t, LLZ_TAG_MDCT) || !d_src || !d_dst || d_src == d_dst) return LLZ_ERR_ARG;
    mdct1_t *f = (mdct1_t *)handle;
    const int prev = llzs_device_enter(f->device);
    const int rc = mdct1_on_device(f, d_src, d_dst, inverse);
    llzs_device_leave(prev);
    return rc;
}

/* the reference's symbols: a frame travels host -> device once and back once */
static void mdct1_run(unsigned long handle, const double *src, double *dst, int inverse, const char *who)
{
    if (!LLZ_HANDLE_OK(handle, mdct1_t, LLZ_TAG_MDCT) || !src || !dst) {
        llzs_set_error("%s: bad handle or arguments", who);
        return;                                                     /* void in the reference ABI */
    }
    mdct1_t *f = (mdct1_t *)handle;
    const int N = f->length, K = N >> 1;
    const int n_src = inverse ? K : N, n_dst = inverse ? N : K;
    double *d_src = inverse ? f->d_bins : f->d_frame, *d_dst = inverse ? f->d_frame : f->d_bins;
    const int prev = llzs_device_enter(f->device);
    int rc = llzs_h2d(d_src, src, sizeof(double) * (size_t)n_src, NULL);
    if (rc == LLZ_OK) rc = mdct1_on_device(f, d_src, d_dst, inverse);
    if (rc == LLZ_OK) rc = llzs_d2h(dst, d_dst, sizeof(double) * (size_t)n_dst, NULL);
    llzs_device_leave(prev);
}

void llz_mdct(unsigned long handle, double *x, double *X) { mdct1_run(handle, x, X, 0, "llz_mdct"); }
void llz_imdct(unsigned long handle, double *X, double *x) { mdct1_run(handle, X, x, 1, "llz_imdct"); }

/* ---- Part 2: batch extension ---- */

typedef struct {
    int tag, length, device;
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
        f->tag = LLZ_TAG_MDCB; f->length = len; f->device = llzs_device_get();
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
        const int prev = llzs_device_enter(((mdcb_t *)handle)->device);
        llzs_sync(((mdcb_t *)handle)->stream);
        mdcb_destroy((mdcb_t *)handle);
        llzs_device_leave(prev);
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
    const int prev = llzs_device_enter(f->device);
    const size_t full = sizeof(float) * (size_t)count * f->length, half = full / 2;
    const size_t ib = inverse ? half : full, ob = inverse ? full : half;
    const int in_dev = llzs_is_device_ptr(in), out_dev = llzs_is_device_ptr(out);
    const float *d_in = in;
    float *d_out = out;
    int rc = (in_dev < 0 || out_dev < 0) ? LLZ_ERR_ARG : LLZ_OK;
    if (rc == LLZ_OK && !in_dev) {
        d_in = (const float *)llz_stage_reserve(&f->st_in, ib);
        rc = d_in ? llzs_h2d((void *)d_in, in, ib, f->stream) : LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK && !out_dev) {
        d_out = (float *)llz_stage_reserve(&f->st_out, ob);
        if (!d_out) rc = LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK) rc = llzs_mdct4_f32(d_in, d_out, count, f->length, f->d_tc, f->d_ts, f->d_cs, inverse, f->stream);
    if (rc == LLZ_OK && !out_dev) rc = llzs_d2h(out, d_out, ob, f->stream);
    llzs_device_leave(prev);
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
