/*
 * llz_mdct_fixed_host.c -- handle layer of the fixed-point MDCT (fixed half): the reference's
 * symbols (reference libllzfilter/llz_mdct_fixed.c:116-526).  Statement order and macros are the reference's; int32
 * additions are written on unsigned operands so that the wrap-around the reference relies on is defined behaviour.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../../include/llz_mdct_fixed.h"
#include "../../../include/llz_fft_fixed.h"
#include "llz_host.h"

#define LLZ_TAG_MDCX 0x4c5a4d58

/* LLZ_FIX15 (llz_fft_fixed.h:42-66): round half away from zero of v * 2^15, saturate to int32, clip to +-32767 */
static short fix15(double v)
{
    const double t = v * (double)(1 << 15);
    const double r = (t > 0) ? floor(t + 0.5) : ceil(t - 0.5);
    int q = r > 2147483647.0 ? 2147483647 : r < -2147483648.0 ? (-2147483647 - 1) : (int)r;
    if (q > 32767) q = 32767;
    if (q < -32767) q = -32767;
    return (short)q;
}
/* LLZ_FIXMUL_32X15 (llz_fft_fixed.h:67) */
static inline int fixmul(int a, int b) { return (int)(((int64_t)a * (int64_t)b) >> 15); }
static inline int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
static inline int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
static inline int wneg(int a) { return (int)(0u - (unsigned)a); }
static inline int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

typedef struct {
    int tag, type, length;
    unsigned long h_fft;
    int *fft_buf, *rot;
    short *d_cos_pos, *d_cos_inv;           /* type 0: Q15 cosine matrices on the device */
    int *d_x, *d_y;
    short *pre_c_pos, *pre_s_pos, *c_pos, *s_pos, *pre_c_inv, *pre_s_inv, *c_inv, *s_inv;
    short *tw_c, *tw_s, sqrt_cof;
} mdcx_t;

static void mdcx_destroy(mdcx_t *f)
{
    if (!f) return;
    if (f->h_fft && f->h_fft != LLZ_BAD_HANDLE) llz_fft_fixed_uninit(f->h_fft);
    free(f->fft_buf); free(f->rot);
    llzs_free(f->d_cos_pos); llzs_free(f->d_cos_inv); llzs_free(f->d_x); llzs_free(f->d_y);
    free(f->pre_c_pos); free(f->pre_s_pos); free(f->c_pos); free(f->s_pos);
    free(f->pre_c_inv); free(f->pre_s_inv); free(f->c_inv); free(f->s_inv);
    free(f->tw_c); free(f->tw_s);
    f->tag = 0;
    free(f);
}

unsigned long llz_mdct_fixed_init(int type, int size)
{
    if (size < 4 || (type != MDCT_FIXED_ORIGIN && type != MDCT_FIXED_FFT && type != MDCT_FIXED_FFT4)) {
        llzs_set_error("llz_mdct_fixed_init: type %d len %d", type, size);
        return LLZ_BAD_HANDLE;
    }
    int base = (int)(log(size) / log(2));                           /* llz_mdct_fixed.c:296-300 */
    if ((1 << base) < size) base += 1;
    const int length = 1 << base;
    const int limit = type == MDCT_FIXED_ORIGIN ? 2048 : (type == MDCT_FIXED_FFT ? 4096 : 16384);
    if (length > limit || (type == MDCT_FIXED_FFT4 && length < 8)) {
        llzs_set_error("llz_mdct_fixed_init: length %d out of range for type %d (at most %d)", length, type, limit);
        return LLZ_BAD_HANDLE;
    }
    mdcx_t *f = (mdcx_t *)calloc(1, sizeof(*f));
    if (!f) return LLZ_BAD_HANDLE;
    f->tag = LLZ_TAG_MDCX; f->type = type; f->length = length;
    int rc = LLZ_OK;
    if (type == MDCT_FIXED_ORIGIN) {                                /* llz_mdct_fixed.c:307-325 */
        const size_t cnt = (size_t)(length >> 1) * length;
        short *pos = (short *)malloc(sizeof(short) * cnt), *inv = (short *)malloc(sizeof(short) * cnt);
        f->d_cos_pos = (short *)llzs_malloc(sizeof(short) * cnt);
        f->d_cos_inv = (short *)llzs_malloc(sizeof(short) * cnt);
        f->d_x = (int *)llzs_malloc(sizeof(int) * (size_t)length);
        f->d_y = (int *)llzs_malloc(sizeof(int) * (size_t)length);
        if (!pos || !inv || !f->d_cos_pos || !f->d_cos_inv || !f->d_x || !f->d_y) rc = LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) {
            for (int k = 0; k < (length >> 1); k++)
                for (int n = 0; n < length; n++) {
                    const double tmp = (M_PI / (2 * length)) * (2 * n + 1 + (length >> 1)) * (2 * k + 1);
                    pos[(size_t)k * length + n] = inv[(size_t)n * (length >> 1) + k] = fix15(cos(tmp));
                }
            rc = llzs_h2d(f->d_cos_pos, pos, sizeof(short) * cnt, NULL);
            if (rc == LLZ_OK) rc = llzs_h2d(f->d_cos_inv, inv, sizeof(short) * cnt, NULL);
        }
        free(pos); free(inv);
    } else if (type == MDCT_FIXED_FFT) {                            /* llz_mdct_fixed.c:326-367 */
        const double n0 = ((double)length / 2 + 1) / 2;
        f->h_fft = llz_fft_fixed_init(length);
        f->fft_buf = (int *)malloc(sizeof(int) * (size_t)length * 2);
        f->pre_c_pos = (short *)malloc(sizeof(short) * (size_t)length);
        f->pre_s_pos = (short *)malloc(sizeof(short) * (size_t)length);
        f->c_pos = (short *)malloc(sizeof(short) * (size_t)(length >> 1));
        f->s_pos = (short *)malloc(sizeof(short) * (size_t)(length >> 1));
        f->pre_c_inv = (short *)malloc(sizeof(short) * (size_t)length);
        f->pre_s_inv = (short *)malloc(sizeof(short) * (size_t)length);
        f->c_inv = (short *)malloc(sizeof(short) * (size_t)length);
        f->s_inv = (short *)malloc(sizeof(short) * (size_t)length);
        if (f->h_fft == LLZ_BAD_HANDLE || !f->fft_buf || !f->pre_c_pos || !f->pre_s_pos || !f->c_pos || !f->s_pos ||
            !f->pre_c_inv || !f->pre_s_inv || !f->c_inv || !f->s_inv) rc = LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) {
            for (int k = 0; k < length; k++) {
                f->pre_c_pos[k] = fix15(cos(-(M_PI * k) / length));
                f->pre_s_pos[k] = fix15(sin(-(M_PI * k) / length));
            }
            for (int k = 0; k < (length >> 1); k++) {
                f->c_pos[k] = fix15(cos(-2 * M_PI * n0 * (k + 0.5) / length));
                f->s_pos[k] = fix15(sin(-2 * M_PI * n0 * (k + 0.5) / length));
            }
            for (int k = 0; k < length; k++) {
                f->pre_c_inv[k] = fix15(cos((2 * M_PI * k * n0) / length));
                f->pre_s_inv[k] = fix15(sin((2 * M_PI * k * n0) / length));
            }
            for (int k = 0; k < length; k++) {
                f->c_inv[k] = fix15(cos(M_PI * (k + n0) / length));
                f->s_inv[k] = fix15(sin(M_PI * (k + n0) / length));
            }
        }
    } else {                                                        /* llz_mdct_fixed.c:368-387 */
        f->h_fft = llz_fft_fixed_init(length >> 2);
        f->fft_buf = (int *)malloc(sizeof(int) * (size_t)(length >> 1));
        f->sqrt_cof = fix15(1. / sqrt(length));
        f->rot = (int *)calloc((size_t)length, sizeof(int));
        f->tw_c = (short *)malloc(sizeof(short) * (size_t)(length >> 2));
        f->tw_s = (short *)malloc(sizeof(short) * (size_t)(length >> 2));
        if (f->h_fft == LLZ_BAD_HANDLE || !f->fft_buf || !f->rot || !f->tw_c || !f->tw_s) rc = LLZ_ERR_NOMEM;
        if (rc == LLZ_OK)
            for (int k = 0; k < (length >> 2); k++) {
                f->tw_c[k] = fix15(cos(-2 * M_PI * (k + 0.125) / length));
                f->tw_s[k] = fix15(sin(-2 * M_PI * (k + 0.125) / length));
            }
    }
    if (rc != LLZ_OK) {
        mdcx_destroy(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

void llz_mdct_fixed_uninit(unsigned long handle)
{
    if (LLZ_HANDLE_OK(handle, mdcx_t, LLZ_TAG_MDCX)) mdcx_destroy((mdcx_t *)handle);
}

static int mdcx_sums(mdcx_t *f, const short *d_A, const int *x, int *y, int rows, int cols)
{
    int rc = llzs_h2d(f->d_x, x, sizeof(int) * (size_t)cols, NULL);
    if (rc == LLZ_OK) rc = llzs_matvec_q15(d_A, f->d_x, f->d_y, rows, cols, NULL);
    if (rc == LLZ_OK) rc = llzs_d2h(y, f->d_y, sizeof(int) * (size_t)rows, NULL);
    return rc;
}

void llz_mdct_fixed(unsigned long handle, int *x, int *X)
{
    if (!LLZ_HANDLE_OK(handle, mdcx_t, LLZ_TAG_MDCX) || !x || !X) {
        llzs_set_error("llz_mdct_fixed: bad handle or arguments");
        return;
    }
    mdcx_t *f = (mdcx_t *)handle;
    const int N = f->length, N2 = N >> 1, N4 = N >> 2;
    if (f->type == MDCT_FIXED_ORIGIN) {                             /* llz_mdct_fixed.c:116-133 */
        (void)mdcx_sums(f, f->d_cos_pos, x, X, N2, N);
    } else if (f->type == MDCT_FIXED_FFT) {                         /* llz_mdct_fixed.c:155-172 */
        for (int k = 0; k < N; k++) {
            f->fft_buf[k + k] = fixmul(x[k], f->pre_c_pos[k]);
            f->fft_buf[k + k + 1] = fixmul(x[k], f->pre_s_pos[k]);
        }
        llz_fft_fixed(f->h_fft, f->fft_buf);
        for (int k = 0; k < N2; k++)
            X[k] = wsub(fixmul(f->fft_buf[k + k], f->c_pos[k]), fixmul(f->fft_buf[k + k + 1], f->s_pos[k]));
    } else {                                                        /* llz_mdct_fixed.c:197-233 */
        int *rot = f->rot;
        memset(rot, 0, sizeof(int) * (size_t)f->length);
        for (int k = 0; k < N4; k++) rot[k] = wneg(x[k + 3 * N4]);
        for (int k = N4; k < N; k++) rot[k] = x[k - N4];
        for (int k = 0; k < N4; k++) {
            const int re = wsub(rot[2 * k], rot[N - 1 - 2 * k]);
            const int im = wsub(rot[N2 - 1 - 2 * k], rot[N2 + 2 * k]);
            f->fft_buf[k + k] = wsub(fixmul(re, f->tw_c[k]), fixmul(im, f->tw_s[k])) >> 1;
            f->fft_buf[k + k + 1] = wadd(fixmul(re, f->tw_s[k]), fixmul(im, f->tw_c[k])) >> 1;
        }
        llz_fft_fixed(f->h_fft, f->fft_buf);
        for (int k = 0; k < N4; k++) {
            const int re = f->fft_buf[k + k], im = f->fft_buf[k + k + 1];
            X[2 * k] = wmul(2, wsub(fixmul(re, f->tw_c[k]), fixmul(im, f->tw_s[k])));
            X[N2 - 1 - 2 * k] = wmul(-2, wadd(fixmul(re, f->tw_s[k]), fixmul(im, f->tw_c[k])));
        }
    }
}

void llz_imdct_fixed(unsigned long handle, int *X, int *x)
{
    if (!LLZ_HANDLE_OK(handle, mdcx_t, LLZ_TAG_MDCX) || !x || !X) {
        llzs_set_error("llz_imdct_fixed: bad handle or arguments");
        return;
    }
    mdcx_t *f = (mdcx_t *)handle;
    const int N = f->length, N2 = N >> 1, N4 = N >> 2;
    if (f->type == MDCT_FIXED_ORIGIN) {                             /* llz_mdct_fixed.c:135-152 */
        if (mdcx_sums(f, f->d_cos_inv, X, x, N, N2) == LLZ_OK)
            for (int n = 0; n < N; n++) x[n] = wmul(x[n], 4) / N;
    } else if (f->type == MDCT_FIXED_FFT) {                         /* llz_mdct_fixed.c:174-195 */
        for (int k = 0; k < N2; k++) {
            f->fft_buf[k + k] = fixmul(X[k], f->pre_c_inv[k]);
            f->fft_buf[k + k + 1] = fixmul(X[k], f->pre_s_inv[k]);
        }
        for (int k = N2, i = N2 - 1; k < N; k++, i--) {
            f->fft_buf[k + k] = fixmul(wneg(X[i]), f->pre_c_inv[k]);
            f->fft_buf[k + k + 1] = fixmul(wneg(X[i]), f->pre_s_inv[k]);
        }
        llz_ifft_fixed(f->h_fft, f->fft_buf);
        for (int k = 0; k < N; k++)
            x[k] = (int)((unsigned)wsub(fixmul(f->fft_buf[k + k], f->c_inv[k]),
                                        fixmul(f->fft_buf[k + k + 1], f->s_inv[k])) << 1);
    } else {                                                        /* llz_mdct_fixed.c:235-283 */
        int *rot = f->rot;
        const short cof = f->sqrt_cof;
        memset(rot, 0, sizeof(int) * (size_t)f->length);
        for (int k = 0; k < N4; k++) {
            const int re = X[2 * k], im = X[N2 - 1 - 2 * k];
            f->fft_buf[k + k] = wsub(fixmul(re, f->tw_c[k]), fixmul(im, f->tw_s[k])) >> 1;
            f->fft_buf[k + k + 1] = wadd(fixmul(re, f->tw_s[k]), fixmul(im, f->tw_c[k])) >> 1;
        }
        llz_fft_fixed(f->h_fft, f->fft_buf);
        for (int k = 0; k < N4; k++) {
            const int re = f->fft_buf[k + k], im = f->fft_buf[k + k + 1];
            int tmp = wsub(fixmul(re, f->tw_c[k]), fixmul(im, f->tw_s[k]));
            f->fft_buf[k + k] = wmul(8, fixmul(tmp, cof));
            tmp = wadd(fixmul(re, f->tw_s[k]), fixmul(im, f->tw_c[k]));
            f->fft_buf[k + k + 1] = wmul(8, fixmul(tmp, cof));
        }
        for (int k = 0; k < N4; k++) {
            rot[2 * k] = f->fft_buf[k + k];
            rot[N2 + 2 * k] = f->fft_buf[k + k + 1];
        }
        for (int k = 1; k < N; k += 2) rot[k] = wneg(rot[N - 1 - k]);
        for (int k = 0; k < 3 * N4; k++) x[k] = fixmul(rot[N4 + k], cof);
        for (int k = 3 * N4; k < N; k++) x[k] = fixmul(wneg(rot[k - 3 * N4]), cof);
    }
}
