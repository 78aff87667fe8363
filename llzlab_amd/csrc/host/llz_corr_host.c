/*
 * llz_corr_host.c -- handle layer of the correlation functions: the reference's symbols
 * (reference libllzfilter/llz_corr.c:38-177) and the float32 batch extension.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../../include/llz_corr.h"
#include "../../../include/llz_fft.h"
#include "llz_host.h"

/* ---- Part 1: reference symbols, exact order on the device ---- */

static void corr_exact(const double *x, const double *y, int n, int p, double *r, const char *who)
{
    if (!x || !y || !r || n < 1 || p < 0) {
        llzs_set_error("%s: bad arguments", who);
        return;                                                     /* void in the reference ABI */
    }
    const size_t nb = sizeof(double) * (size_t)n, rb = sizeof(double) * ((size_t)p + 1);
    double *d_x = (double *)llzs_malloc(nb);
    double *d_y = (x == y) ? d_x : (double *)llzs_malloc(nb);
    double *d_r = (double *)llzs_malloc(rb);
    if (d_x && d_y && d_r && llzs_h2d(d_x, x, nb, NULL) == LLZ_OK &&
        (x == y || llzs_h2d(d_y, y, nb, NULL) == LLZ_OK) &&
        llzs_corr_exact_f64(d_x, d_y, n, p, d_r, NULL) == LLZ_OK)
        (void)llzs_d2h(r, d_r, rb, NULL);
    if (d_y != d_x) llzs_free(d_y);
    llzs_free(d_x);
    llzs_free(d_r);
}

void llz_autocorr(double *x, int n, int p, double *r) { corr_exact(x, x, n, p, r, "llz_autocorr"); }

void llz_crosscorr(double *x, double *y, int n, int p, double *r) { corr_exact(x, y, n, p, r, "llz_crosscorr"); }

double llz_corr_cof(double *a, double *b, int len)
{
    /* the three running sums of llz_corr.c:70-74 are lag-0 correlations; the final division and square root run on
     * the host's libm exactly as in the reference */
    double ab = 0, aa = 0, bb = 0;
    corr_exact(a, b, len, 0, &ab, "llz_corr_cof");
    corr_exact(a, a, len, 0, &aa, "llz_corr_cof");
    corr_exact(b, b, len, 0, &bb, "llz_corr_cof");
    return ab / sqrt(aa * bb);
}

typedef struct {
    int tag, n, fft_len;
    unsigned long h_fft;
    double *b1, *b2;
} acf1_t;

#define LLZ_TAG_ACF1 0x4c5a4131
#define LLZ_TAG_ACFM 0x4c5a414d

static int acf_fft_len(int n)
{
    int level = (int)log2((double)(2 * n));                        /* llz_corr.c:81-93, 106 */
    if ((1 << level) < 2 * n) level += 1;
    return 1 << level;
}

unsigned long llz_autocorr_fast_init(int n)
{
    if (n < 1 || acf_fft_len(n) > 4096) {
        llzs_set_error("llz_autocorr_fast_init: n=%d (1..2048)", n);
        return LLZ_BAD_HANDLE;
    }
    acf1_t *f = (acf1_t *)calloc(1, sizeof(*f));
    if (!f) return LLZ_BAD_HANDLE;
    f->tag = LLZ_TAG_ACF1; f->n = n; f->fft_len = acf_fft_len(n);
    f->h_fft = llz_fft_init(f->fft_len);
    f->b1 = (double *)calloc(2 * (size_t)f->fft_len, sizeof(double));
    f->b2 = (double *)calloc(2 * (size_t)f->fft_len, sizeof(double));
    if (f->h_fft == LLZ_BAD_HANDLE || !f->b1 || !f->b2) {
        if (f->h_fft != LLZ_BAD_HANDLE) llz_fft_uninit(f->h_fft);
        free(f->b1); free(f->b2); free(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

void llz_autocorr_fast_uninit(unsigned long handle)
{
    if (!LLZ_HANDLE_OK(handle, acf1_t, LLZ_TAG_ACF1)) return;
    acf1_t *f = (acf1_t *)handle;
    llz_fft_uninit(f->h_fft);
    free(f->b1); free(f->b2);
    f->tag = 0;
    free(f);
}

void llz_autocorr_fast(unsigned long handle, double *x, int n, int p, double *r)
{
    if (!LLZ_HANDLE_OK(handle, acf1_t, LLZ_TAG_ACF1) || !x || !r || n < 1 || p < 0) {
        llzs_set_error("llz_autocorr_fast: bad handle or arguments");
        return;
    }
    acf1_t *f = (acf1_t *)handle;
    if (n > f->fft_len / 2 || p >= f->fft_len) {
        llzs_set_error("llz_autocorr_fast: n=%d / p=%d do not fit fft_len %d", n, p, f->fft_len);
        return;
    }
    /* llz_corr.c:155-177 with both transforms on the device (exact-order double FFT): same bits as the reference */
    memset(f->b1, 0, sizeof(double) * 2 * (size_t)f->fft_len);
    for (int i = 0; i < n; i++) f->b1[2 * i] = x[i];
    llz_fft(f->h_fft, f->b1);
    memset(f->b2, 0, sizeof(double) * 2 * (size_t)f->fft_len);
    for (int i = 0; i < n; i++)                                     /* only the first n bins: the reference's quirk */
        f->b2[2 * i] = f->b1[2 * i] * f->b1[2 * i] + f->b1[2 * i + 1] * f->b1[2 * i + 1];
    llz_ifft(f->h_fft, f->b2);
    for (int i = 0; i <= p; i++) r[i] = f->b2[2 * i] * 2;
}

/* ---- Part 2: batch extension ---- */

int llz_autocorr_mc(const float *x, float *r, int frames, int n, int p, void *stream)
{
    if (!x || !r || frames < 1 || n < 1 || p < 0 || p >= n || p > 255) {
        llzs_set_error("llz_autocorr_mc: frames %d n %d p %d (p < n, p <= 255)", frames, n, p);
        return LLZ_ERR_ARG;
    }
    const size_t xb = sizeof(float) * (size_t)frames * n, rb = sizeof(float) * (size_t)frames * (p + 1);
    const int x_dev = llzs_is_device_ptr(x), r_dev = llzs_is_device_ptr(r);
    if (x_dev < 0 || r_dev < 0) return LLZ_ERR_ARG;               /* device memory of a GPU that is not current */
    float *d_x = (float *)x, *d_r = r;
    int rc = LLZ_OK;
    if (!x_dev) {
        d_x = (float *)llzs_malloc(xb);
        rc = d_x ? llzs_h2d(d_x, x, xb, stream) : LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK && !r_dev) {
        d_r = (float *)llzs_malloc(rb);
        if (!d_r) rc = LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK) rc = llzs_autocorr_mc_f32(d_x, d_r, frames, n, p, stream);
    if (rc == LLZ_OK && !r_dev) rc = llzs_d2h(r, d_r, rb, stream);
    if (!x_dev) { llzs_sync(stream); llzs_free(d_x); }
    if (!r_dev) llzs_free(d_r);
    return rc;
}

typedef struct {
    int tag, device, frames, n, fft_len;
    float *d_cs;            /* fft_len cos then fft_len sin */
    void *stream;
    llz_stage_t st_in, st_out;
} acfm_t;

static void acfm_destroy(acfm_t *f)
{
    if (!f) return;
    llzs_free(f->d_cs);
    llz_stage_release(&f->st_in); llz_stage_release(&f->st_out);
    f->tag = 0;
    free(f);
}

unsigned long llz_autocorr_fast_mc_init(int frames, int n)
{
    if (frames < 1 || n < 1 || acf_fft_len(n) > 4096 || acf_fft_len(n) < 8) {
        llzs_set_error("llz_autocorr_fast_mc_init: frames %d n %d (4..2048)", frames, n);
        return LLZ_BAD_HANDLE;
    }
    acfm_t *f = (acfm_t *)calloc(1, sizeof(*f));
    if (!f) return LLZ_BAD_HANDLE;
    f->tag = LLZ_TAG_ACFM; f->device = llzs_device_get(); f->frames = frames; f->n = n; f->fft_len = acf_fft_len(n);
    const int F = f->fft_len;
    float *cs = (float *)malloc(sizeof(float) * 2 * (size_t)F);
    int rc = cs ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK) {
        for (int i = 0; i < F; i++) {
            const double ang = (double)(2 * M_PI * i) / F;
            cs[i] = (float)cos(ang);
            cs[F + i] = (float)sin(ang);
        }
        f->d_cs = (float *)llzs_malloc(sizeof(float) * 2 * (size_t)F);
        rc = f->d_cs ? llzs_h2d(f->d_cs, cs, sizeof(float) * 2 * (size_t)F, NULL) : LLZ_ERR_NOMEM;
    }
    free(cs);
    if (rc != LLZ_OK) {
        acfm_destroy(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

void llz_autocorr_fast_mc_uninit(unsigned long handle)
{
    if (LLZ_HANDLE_OK(handle, acfm_t, LLZ_TAG_ACFM)) {
        const int prev = llzs_device_enter(((acfm_t *)handle)->device);
        llzs_sync(((acfm_t *)handle)->stream);
        acfm_destroy((acfm_t *)handle);
        llzs_device_leave(prev);
    }
}

int llz_autocorr_fast_mc_set_stream(unsigned long handle, void *stream)
{
    if (!LLZ_HANDLE_OK(handle, acfm_t, LLZ_TAG_ACFM)) return LLZ_ERR_ARG;
    ((acfm_t *)handle)->stream = stream;
    return LLZ_OK;
}

int llz_autocorr_fast_mc(unsigned long handle, const float *x, float *r, int p)
{
    if (!LLZ_HANDLE_OK(handle, acfm_t, LLZ_TAG_ACFM) || !x || !r || p < 0) {
        llzs_set_error("llz_autocorr_fast_mc: bad handle or arguments");
        return LLZ_ERR_ARG;
    }
    acfm_t *f = (acfm_t *)handle;
    if (p >= f->fft_len) {
        llzs_set_error("llz_autocorr_fast_mc: p=%d >= fft_len %d", p, f->fft_len);
        return LLZ_ERR_ARG;
    }
    const size_t xb = sizeof(float) * (size_t)f->frames * f->n, rb = sizeof(float) * (size_t)f->frames * (p + 1);
    const int prev = llzs_device_enter(f->device);
    const int x_dev = llzs_is_device_ptr(x), r_dev = llzs_is_device_ptr(r);
    const float *d_x = x;
    float *d_r = r;
    int rc = (x_dev < 0 || r_dev < 0) ? LLZ_ERR_ARG : LLZ_OK;     /* a buffer of another GPU: refused, message set */
    if (rc == LLZ_OK && !x_dev) {
        d_x = (const float *)llz_stage_reserve(&f->st_in, xb);
        rc = d_x ? llzs_h2d((void *)d_x, x, xb, f->stream) : LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK && !r_dev) {
        d_r = (float *)llz_stage_reserve(&f->st_out, rb);
        if (!d_r) rc = LLZ_ERR_NOMEM;
    }
    /* pack -> FFT -> |X|^2 (first n bins) -> IFFT -> 2 Re, fused in LDS */
    if (rc == LLZ_OK) rc = llzs_acf_fused_f32(d_x, d_r, f->frames, f->n, p, f->fft_len, f->d_cs, f->stream);
    if (rc == LLZ_OK && !r_dev) rc = llzs_d2h(r, d_r, rb, f->stream);
    llzs_device_leave(prev);
    return rc;
}
