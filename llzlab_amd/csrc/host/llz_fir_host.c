/*
 * llz_fir_host.c -- handle layer of the FIR path: the reference's single-channel `double` symbols
 * (reference libllzfilter/llz_fir.c:442-625) and the multi-channel float32 batch extension.  Plain C; the device
 * is reached only through llz_shim.h.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../../include/llz_fir.h"
#include "llz_host.h"

/* =====================================================================================================
 * Part 1: single channel, double, bit-exact with the reference (exact-order kernel k_fir_td_f64_exact)
 * ===================================================================================================== */

typedef struct {
    int tag;
    int flt_len, frame_len;
    double *h;          /* host taps */
    double *xbuf;       /* host: [flt_len-1 history | frame_len samples] */
    double *d_taps;     /* device taps */
    double *d_x;        /* device: same layout as xbuf */
    double *d_y;        /* device: frame_len outputs */
} fir1_t;

static void fir1_destroy(fir1_t *f)
{
    if (!f) return;
    free(f->h); free(f->xbuf);
    llzs_free(f->d_taps); llzs_free(f->d_x); llzs_free(f->d_y);
    f->tag = 0;
    free(f);
}

static unsigned long fir1_create(int kind, int frame_len, int flt_len, double fc1, double fc2, win_t win)
{
    if (frame_len < 1 || flt_len < 1) {
        llzs_set_error("llz_fir_filter_*_init: frame_len %d / flt_len %d", frame_len, flt_len);
        return LLZ_BAD_HANDLE;
    }
    fir1_t *f = (fir1_t *)calloc(1, sizeof(*f));
    if (!f) return LLZ_BAD_HANDLE;
    f->tag = LLZ_TAG_FIR1;
    f->frame_len = frame_len;
    f->flt_len = llz_host_design(kind, &f->h, flt_len, fc1, fc2, win);    /* stored length = returned length */
    if (f->flt_len < 1) {
        llzs_set_error("llz_fir_filter_*_init: tap design failed (win_type %d)", win);
        fir1_destroy(f);
        return LLZ_BAD_HANDLE;
    }
    /* flush feeds flt_len-1 zeros, so the frame area must hold max(frame_len, flt_len-1) samples */
    const int keep = f->flt_len - 1;
    const int span = keep + (frame_len > keep ? frame_len : keep) + 1;
    f->xbuf = (double *)calloc((size_t)span, sizeof(double));            /* zero history: llz_fir.c:459 */
    f->d_taps = (double *)llzs_malloc(sizeof(double) * (size_t)f->flt_len);
    f->d_x = (double *)llzs_malloc(sizeof(double) * (size_t)span);
    f->d_y = (double *)llzs_malloc(sizeof(double) * (size_t)(span - keep));
    if (!f->xbuf || !f->d_taps || !f->d_x || !f->d_y ||
        llzs_h2d(f->d_taps, f->h, sizeof(double) * (size_t)f->flt_len, NULL) != LLZ_OK) {
        fir1_destroy(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

unsigned long llz_fir_filter_lpf_init(int frame_len, int flt_len, double fc, win_t win_type)
{
    return fir1_create(LLZ_KIND_LPF, frame_len, flt_len, fc, 0.0, win_type);
}

unsigned long llz_fir_filter_hpf_init(int frame_len, int flt_len, double fc, win_t win_type)
{
    return fir1_create(LLZ_KIND_HPF, frame_len, flt_len, fc, 0.0, win_type);
}

unsigned long llz_fir_filter_bandpass_init(int frame_len, int flt_len, double fc1, double fc2, win_t win_type)
{
    return fir1_create(LLZ_KIND_BPF, frame_len, flt_len, fc1, fc2, win_type);
}

unsigned long llz_fir_filter_bandstop_init(int frame_len, int flt_len, double fc1, double fc2, win_t win_type)
{
    return fir1_create(LLZ_KIND_BSF, frame_len, flt_len, fc1, fc2, win_type);
}

void llz_fir_filter_uninit(unsigned long handle)
{
    if (LLZ_HANDLE_OK(handle, fir1_t, LLZ_TAG_FIR1))
        fir1_destroy((fir1_t *)handle);
}

/* run `count` samples sitting behind the history in xbuf through the device; copy `emit` results out */
static int fir1_run(fir1_t *f, int count, double *dst, int emit)
{
    const int keep = f->flt_len - 1;
    const int span = count > emit ? count : emit;
    int rc = llzs_h2d(f->d_x, f->xbuf, sizeof(double) * (size_t)(keep + span), NULL);
    if (rc == LLZ_OK)
        rc = llzs_fir_td_f64(f->d_x + keep, f->d_y, f->d_x, f->d_taps, emit, f->flt_len, NULL);
    if (rc == LLZ_OK)
        rc = llzs_d2h(dst, f->d_y, sizeof(double) * (size_t)emit, NULL);
    /* new history = last keep samples of [history | the init frame length] (llz_fir.c:562-564) */
    memmove(f->xbuf, f->xbuf + count, sizeof(double) * (size_t)keep);
    return rc;
}

int llz_fir_filter(unsigned long handle, double *buf_in, double *buf_out, int frame_len)
{
    if (!LLZ_HANDLE_OK(handle, fir1_t, LLZ_TAG_FIR1) || !buf_in || !buf_out) {
        llzs_set_error("llz_fir_filter: bad handle or NULL buffer");
        return LLZ_ERR_ARG;
    }
    fir1_t *f = (fir1_t *)handle;
    if (frame_len != f->frame_len) {
        /* the reference asserts frame_len <= init and mis-shifts its history for anything shorter */
        llzs_set_error("llz_fir_filter: frame_len %d != init frame_len %d", frame_len, f->frame_len);
        return LLZ_ERR_ARG;
    }
    memcpy(f->xbuf + (f->flt_len - 1), buf_in, sizeof(double) * (size_t)frame_len);
    const int rc = fir1_run(f, frame_len, buf_out, frame_len);
    return rc == LLZ_OK ? frame_len : rc;
}

int llz_fir_filter_flush(unsigned long handle, double *buf_out)
{
    if (!LLZ_HANDLE_OK(handle, fir1_t, LLZ_TAG_FIR1) || !buf_out) {
        llzs_set_error("llz_fir_filter_flush: bad handle or NULL buffer");
        return LLZ_ERR_ARG;
    }
    fir1_t *f = (fir1_t *)handle;
    const int keep = f->flt_len - 1;
    if (keep == 0) return 0;
    /* llz_fir.c:608-622: a frame of zeros goes in, the first flt_len-1 outputs come out.  (The reference reads
     * past its buffer when flt_len-2 >= frame_len; here the frame area is always large enough and zeroed.) */
    const int zeros = f->frame_len > keep ? f->frame_len : keep;
    memset(f->xbuf + keep, 0, sizeof(double) * (size_t)zeros);
    const int rc = fir1_run(f, f->frame_len, buf_out, keep);
    return rc == LLZ_OK ? keep : rc;
}

/* =====================================================================================================
 * Part 2: multi-channel float32 batch (kernels K1 k_fir_td_f32 and K4 k_fir_ols_f32)
 * ===================================================================================================== */

typedef struct {
    int tag;
    int device;                 /* the device the handle's buffers live on: every call binds it */
    int channels, frame_len, flt_len, algo;
    float *d_taps;              /* flt_len floats zero-padded to a multiple of 16 */
    float *d_hfreq, *d_twid;    /* overlap-save tables (NULL for the time-domain algorithm) */
    float *d_tw2k;              /* 2048-point overlap-save: W_2048^n, n < 1024 (d_hfreq: even | odd bins) */
    float *d_tw4k;              /* 4096-point overlap-save on a whole wave (<= 3073 taps): W_4096^n, n < 2048 (d_hfreq: 4 planes) */
    float *d_hist[2];           /* [channels][flt_len-1], ping-pong */
    int cur;
    float *d_zero;              /* [channels][flt_len-1] zeros: flush input */
    void *stream;
    llz_stage_t st_in, st_out;  /* only for callers passing host memory */
} firm_t;

static void firm_destroy(firm_t *f)
{
    if (!f) return;
    llzs_free(f->d_taps); llzs_free(f->d_hfreq); llzs_free(f->d_twid); llzs_free(f->d_tw2k); llzs_free(f->d_tw4k);
    llzs_free(f->d_hist[0]); llzs_free(f->d_hist[1]); llzs_free(f->d_zero);
    llz_stage_release(&f->st_in); llz_stage_release(&f->st_out);
    f->tag = 0;
    free(f);
}

/* spectrum of the (float-rounded) taps, scaled by 1/N, as float pairs in natural bin order; and the 32x32
 * inter-pass twiddles W_N^(a*b).  Direct DFT in double: 257 x 1024 terms, setup time only. */
static int firm_build_ols_tables(firm_t *f, const float *taps)
{
    const int N = LLZS_OLS_NFFT;
    float *hf = (float *)malloc(sizeof(float) * 2 * (size_t)N);
    float *tw = (float *)malloc(sizeof(float) * 2 * 1024);
    double *cs = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    int rc = LLZ_ERR_NOMEM;
    if (hf && tw && cs) {
        for (int i = 0; i < N; i++) {
            /* exact quadrant values keep the table symmetric */
            const double ang = 2.0 * M_PI * (double)i / (double)N;
            cs[2 * i] = (i == N / 4 || i == 3 * N / 4) ? 0.0 : cos(ang);
            cs[2 * i + 1] = (i == 0 || i == N / 2) ? 0.0 : sin(ang);
        }
        for (int k = 0; k < N; k++) {
            double re = 0.0, im = 0.0;
            for (int t = 0; t < f->flt_len; t++) {
                const int m = (int)(((long)k * t) % N);
                re += (double)taps[t] * cs[2 * m];
                im -= (double)taps[t] * cs[2 * m + 1];
            }
            hf[2 * k] = (float)(re / N);
            hf[2 * k + 1] = (float)(im / N);
        }
        for (int a = 0; a < 32; a++)
            for (int b = 0; b < 32; b++) {
                const int m = (a * b) % N;
                tw[2 * (a * 32 + b)] = (float)cs[2 * m];
                tw[2 * (a * 32 + b) + 1] = (float)(-cs[2 * m + 1]);       /* W = exp(-2 pi j m / N) */
            }
        f->d_hfreq = (float *)llzs_malloc(sizeof(float) * 2 * (size_t)N);
        f->d_twid = (float *)llzs_malloc(sizeof(float) * 2 * 1024);
        rc = (f->d_hfreq && f->d_twid) ? LLZ_OK : LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_hfreq, hf, sizeof(float) * 2 * (size_t)N);
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_twid, tw, sizeof(float) * 2 * 1024);
    }
    free(hf); free(tw); free(cs);
    return rc;
}

/* 258 .. 1025 taps (k_fir_ols2k_chain_f32): DFT_2048(taps) / 2048 as [even bins | odd bins], the 32 x 32 twiddles of the
 * 1024-point halves, and W_2048^n for the radix-2 step.  Direct DFT in double, setup time only. */
static int firm_build_ols2k_tables(firm_t *f, const float *taps)
{
    const int N = 2048, H = 1024;
    float *hf = (float *)malloc(sizeof(float) * 2 * (size_t)N);
    float *tw = (float *)malloc(sizeof(float) * 2 * 1024);
    float *w2 = (float *)malloc(sizeof(float) * 2 * (size_t)H);
    double *cs = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    int rc = LLZ_ERR_NOMEM;
    if (hf && tw && w2 && cs) {
        for (int i = 0; i < N; i++) {
            const double ang = 2.0 * M_PI * (double)i / (double)N;
            cs[2 * i] = (i == N / 4 || i == 3 * N / 4) ? 0.0 : cos(ang);
            cs[2 * i + 1] = (i == 0 || i == N / 2) ? 0.0 : sin(ang);
        }
        for (int k = 0; k < N; k++) {
            double re = 0.0, im = 0.0;
            for (int t = 0; t < f->flt_len; t++) {
                const int m = (int)(((long)k * t) % N);
                re += (double)taps[t] * cs[2 * m];
                im -= (double)taps[t] * cs[2 * m + 1];
            }
            const int dst = (k & 1) * H + (k >> 1);                    /* even bins first, then odd bins */
            hf[2 * dst] = (float)(re / N);
            hf[2 * dst + 1] = (float)(im / N);
        }
        for (int a = 0; a < 32; a++)
            for (int b = 0; b < 32; b++) {
                const int m = (2 * a * b) % N;                         /* W_1024^(ab) = W_2048^(2ab) */
                tw[2 * (a * 32 + b)] = (float)cs[2 * m];
                tw[2 * (a * 32 + b) + 1] = (float)(-cs[2 * m + 1]);
            }
        for (int i = 0; i < H; i++) {
            w2[2 * i] = (float)cs[2 * i];
            w2[2 * i + 1] = (float)(-cs[2 * i + 1]);                   /* W = exp(-2 pi j i / 2048) */
        }
        f->d_hfreq = (float *)llzs_malloc(sizeof(float) * 2 * (size_t)N);
        f->d_twid = (float *)llzs_malloc(sizeof(float) * 2 * 1024);
        f->d_tw2k = (float *)llzs_malloc(sizeof(float) * 2 * (size_t)H);
        rc = (f->d_hfreq && f->d_twid && f->d_tw2k) ? LLZ_OK : LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_hfreq, hf, sizeof(float) * 2 * (size_t)N);
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_twid, tw, sizeof(float) * 2 * 1024);
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_tw2k, w2, sizeof(float) * 2 * (size_t)H);
    }
    free(hf); free(tw); free(w2); free(cs);
    return rc;
}

/* 514 .. 3073 taps (k_fir_ols4k_f32): DFT_4096(taps) / 4096 as four planes (plane j = bins 4m + j), the 32 x 32 twiddles
 * of the 1024-point transforms, W_2048^n and W_4096^n for the two radix-2 steps.  Direct DFT in double, setup time only. */
static int firm_build_ols4k_tables(firm_t *f, const float *taps)
{
    const int N = 4096, Q = 1024;
    float *hf = (float *)malloc(sizeof(float) * 2 * (size_t)N);
    float *tw = (float *)malloc(sizeof(float) * 2 * 1024);
    float *w2 = (float *)malloc(sizeof(float) * 2 * 1024);
    float *w4 = (float *)malloc(sizeof(float) * 2 * 2048);
    double *cs = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    int rc = LLZ_ERR_NOMEM;
    if (hf && tw && w2 && w4 && cs) {
        for (int i = 0; i < N; i++) {
            const double ang = 2.0 * M_PI * (double)i / (double)N;
            cs[2 * i] = (i == N / 4 || i == 3 * N / 4) ? 0.0 : cos(ang);
            cs[2 * i + 1] = (i == 0 || i == N / 2) ? 0.0 : sin(ang);
        }
        for (int k = 0; k < N; k++) {
            double re = 0.0, im = 0.0;
            for (int t = 0; t < f->flt_len; t++) {
                const int m = (int)(((long)k * t) % N);
                re += (double)taps[t] * cs[2 * m];
                im -= (double)taps[t] * cs[2 * m + 1];
            }
            const int dst = (k & 3) * Q + (k >> 2);
            hf[2 * dst] = (float)(re / N);
            hf[2 * dst + 1] = (float)(im / N);
        }
        for (int a = 0; a < 32; a++)
            for (int b = 0; b < 32; b++) {
                const int m = (4 * a * b) % N;                         /* W_1024^(ab) = W_4096^(4ab) */
                tw[2 * (a * 32 + b)] = (float)cs[2 * m];
                tw[2 * (a * 32 + b) + 1] = (float)(-cs[2 * m + 1]);
            }
        for (int i = 0; i < 1024; i++) { w2[2 * i] = (float)cs[2 * (2 * i)]; w2[2 * i + 1] = (float)(-cs[2 * (2 * i) + 1]); }
        for (int i = 0; i < 2048; i++) { w4[2 * i] = (float)cs[2 * i]; w4[2 * i + 1] = (float)(-cs[2 * i + 1]); }
        f->d_hfreq = (float *)llzs_malloc(sizeof(float) * 2 * (size_t)N);
        f->d_twid = (float *)llzs_malloc(sizeof(float) * 2 * 1024);
        f->d_tw2k = (float *)llzs_malloc(sizeof(float) * 2 * 1024);
        f->d_tw4k = (float *)llzs_malloc(sizeof(float) * 2 * 2048);
        rc = (f->d_hfreq && f->d_twid && f->d_tw2k && f->d_tw4k) ? LLZ_OK : LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_hfreq, hf, sizeof(float) * 2 * (size_t)N);
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_twid, tw, sizeof(float) * 2 * 1024);
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_tw2k, w2, sizeof(float) * 2 * 1024);
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_tw4k, w4, sizeof(float) * 2 * 2048);
    }
    free(hf); free(tw); free(w2); free(w4); free(cs);
    return rc;
}

/* 2 .. 6145 taps on pairs of waves (k_fir_ols8k_f32): DFT_8192(taps) / 8192 as eight planes (plane j = bins 8m + j), the
 * 32 x 32 twiddles of the 1024-point transforms and W_4096^n (the kernel forms W_8192^n itself).  Direct DFT in double. */
static int firm_build_ols8k_tables(firm_t *f, const float *taps)
{
    const int N = 8192, Q = 1024;
    float *hf = (float *)malloc(sizeof(float) * 2 * (size_t)N);
    float *tw = (float *)malloc(sizeof(float) * 2 * 1024);
    float *w4 = (float *)malloc(sizeof(float) * 2 * 2048);
    double *cs = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    int rc = LLZ_ERR_NOMEM;
    if (hf && tw && w4 && cs) {
        for (int i = 0; i < N; i++) {
            const double ang = 2.0 * M_PI * (double)i / (double)N;
            cs[2 * i] = (i == N / 4 || i == 3 * N / 4) ? 0.0 : cos(ang);
            cs[2 * i + 1] = (i == 0 || i == N / 2) ? 0.0 : sin(ang);
        }
        for (int k = 0; k < N; k++) {
            double re = 0.0, im = 0.0;
            for (int t = 0; t < f->flt_len; t++) {
                const int m = (int)(((long)k * t) % N);
                re += (double)taps[t] * cs[2 * m];
                im -= (double)taps[t] * cs[2 * m + 1];
            }
            const int dst = (k & 7) * Q + (k >> 3);
            hf[2 * dst] = (float)(re / N);
            hf[2 * dst + 1] = (float)(im / N);
        }
        for (int a = 0; a < 32; a++)
            for (int b = 0; b < 32; b++) {
                const int m = (8 * a * b) % N;                         /* W_1024^(ab) = W_8192^(8ab) */
                tw[2 * (a * 32 + b)] = (float)cs[2 * m];
                tw[2 * (a * 32 + b) + 1] = (float)(-cs[2 * m + 1]);
            }
        for (int i = 0; i < 2048; i++) { w4[2 * i] = (float)cs[2 * (2 * i)]; w4[2 * i + 1] = (float)(-cs[2 * (2 * i) + 1]); }
        f->d_hfreq = (float *)llzs_malloc(sizeof(float) * 2 * (size_t)N);
        f->d_twid = (float *)llzs_malloc(sizeof(float) * 2 * 1024);
        f->d_tw4k = (float *)llzs_malloc(sizeof(float) * 2 * 2048);
        rc = (f->d_hfreq && f->d_twid && f->d_tw4k) ? LLZ_OK : LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_hfreq, hf, sizeof(float) * 2 * (size_t)N);
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_twid, tw, sizeof(float) * 2 * 1024);
        if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_tw4k, w4, sizeof(float) * 2 * 2048);
    }
    free(hf); free(tw); free(w4); free(cs);
    return rc;
}

unsigned long llz_fir_filter_mc_init(int channels, int frame_len, const float *taps, int flt_len, int algo)
{
    if (channels < 1 || channels > 65535 || frame_len < 1 || !taps || flt_len < 1) {
        llzs_set_error("llz_fir_filter_mc_init: channels %d frame_len %d flt_len %d", channels, frame_len, flt_len);
        return LLZ_BAD_HANDLE;
    }
    if (!llzs_fir_td_f32_fits(flt_len)) {
        /* every handle flushes through the time-domain kernel (and AUTO falls back to it): refuse here, not at the first call */
        llzs_set_error("llz_fir_filter_mc_init: %d taps exceed the time-domain kernel's LDS tile", flt_len);
        return LLZ_BAD_HANDLE;
    }
    if (algo == LLZ_FIR_ALGO_AUTO) {
        /* measured on 4096 ch x 2^20 (tools/fir_crossover.py): time domain 6.0 / 6.4 / 7.9 / 9.5 ms at 9 / 17 / 33 / 63
         * taps, overlap-save 6.7 ms at any length up to 257 -> overlap-save from 33 taps on; beyond 257 taps the
         * matrix-core form of the time domain (23.7 ms at 257 taps against 31.9 ms on the VALU) */
        if (flt_len <= 32) algo = LLZ_FIR_ALGO_TIME;
        else if (flt_len <= LLZS_OLS_MAX_TAPS) algo = LLZ_FIR_ALGO_OVERLAP_SAVE;
        /* whole-wave overlap-save (4096 ch x 2^20): 2048 points with 512 of overlap 7.8 ms up to 513 taps (10.6 ms with 1024
         * of overlap); 4096 points 8.0 / 7.9 / 8.3 / 9.6 / 11.3 / 14.7 / 19.5 ms with 512 / 768 / 1024 / 1536 / 2048 / 2560 /
         * 3072 of overlap */
        else if (flt_len <= 513) algo = LLZ_FIR_ALGO_OVERLAP_SAVE_2048;
        /* 8192 points on pairs of waves, same batch: 9.1 / 10.9 / 10.9 / 11.3 / 13.8 ms with 1536 / 2304 / 2560 / 3072 / 4096 of
         * overlap -> from 1026 taps on (4096 points: 9.4 / 11.2 / 14.4 / 19.2 ms with 1536 / 2048 / 2560 / 3072), up to 6145 taps */
        else if (flt_len <= 1025) algo = LLZ_FIR_ALGO_OVERLAP_SAVE_4096;
        else if (flt_len <= LLZS_OLS8K_MAX_TAPS) algo = LLZ_FIR_ALGO_OVERLAP_SAVE_8192;
        else algo = llzs_fir_mfma_f32_fits(flt_len, 1) ? LLZ_FIR_ALGO_TIME_MFMA : LLZ_FIR_ALGO_TIME;
    }
    if (algo == LLZ_FIR_ALGO_OVERLAP_SAVE_2048 && (flt_len < 2 || flt_len > LLZS_OLS2K_MAX_TAPS)) {
        llzs_set_error("llz_fir_filter_mc_init: the 2048-point overlap-save takes 2..%d taps", LLZS_OLS2K_MAX_TAPS);
        return LLZ_BAD_HANDLE;
    }
    if (algo == LLZ_FIR_ALGO_OVERLAP_SAVE_4096 && (flt_len < 2 || flt_len > LLZS_OLS4K_MAX_TAPS)) {
        llzs_set_error("llz_fir_filter_mc_init: the 4096-point overlap-save takes 2..%d taps", LLZS_OLS4K_MAX_TAPS);
        return LLZ_BAD_HANDLE;
    }
    if (algo == LLZ_FIR_ALGO_OVERLAP_SAVE_8192 && (flt_len < 2 || flt_len > LLZS_OLS8K_MAX_TAPS)) {
        llzs_set_error("llz_fir_filter_mc_init: the 8192-point overlap-save takes 2..%d taps", LLZS_OLS8K_MAX_TAPS);
        return LLZ_BAD_HANDLE;
    }
    if (algo == LLZ_FIR_ALGO_OVERLAP_SAVE && flt_len > LLZS_OLS_MAX_TAPS) {
        llzs_set_error("llz_fir_filter_mc_init: overlap-save supports at most %d taps", LLZS_OLS_MAX_TAPS);
        return LLZ_BAD_HANDLE;
    }
    if (algo == LLZ_FIR_ALGO_TIME_MFMA && !llzs_fir_mfma_f32_fits(flt_len, 1)) {
        llzs_set_error("llz_fir_filter_mc_init: %d taps do not fit the matrix-core kernel's LDS tile", flt_len);
        return LLZ_BAD_HANDLE;
    }
    if (algo != LLZ_FIR_ALGO_TIME && algo != LLZ_FIR_ALGO_OVERLAP_SAVE && algo != LLZ_FIR_ALGO_TIME_MFMA &&
        algo != LLZ_FIR_ALGO_OVERLAP_SAVE_2048 && algo != LLZ_FIR_ALGO_OVERLAP_SAVE_4096 && algo != LLZ_FIR_ALGO_OVERLAP_SAVE_8192) {
        llzs_set_error("llz_fir_filter_mc_init: unknown algo %d", algo);
        return LLZ_BAD_HANDLE;
    }
    firm_t *f = (firm_t *)calloc(1, sizeof(*f));
    if (!f) return LLZ_BAD_HANDLE;
    f->tag = LLZ_TAG_FIRM;
    f->device = llzs_device_get();
    f->channels = channels; f->frame_len = frame_len; f->flt_len = flt_len; f->algo = algo;

    const int tpad = (flt_len + 15) & ~15;
    const size_t hist_bytes = sizeof(float) * (size_t)channels * (size_t)(flt_len > 1 ? flt_len - 1 : 1);
    float *padded = (float *)calloc((size_t)tpad, sizeof(float));
    int rc = padded ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK) {
        memcpy(padded, taps, sizeof(float) * (size_t)flt_len);
        f->d_taps = (float *)llzs_malloc(sizeof(float) * (size_t)tpad);
        f->d_hist[0] = (float *)llzs_malloc(hist_bytes);
        f->d_hist[1] = (float *)llzs_malloc(hist_bytes);
        f->d_zero = (float *)llzs_malloc(hist_bytes);
        if (!f->d_taps || !f->d_hist[0] || !f->d_hist[1] || !f->d_zero) rc = LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK) rc = llzs_h2d_table(f->d_taps, padded, sizeof(float) * (size_t)tpad);
    if (rc == LLZ_OK) rc = llzs_memset(f->d_hist[0], 0, hist_bytes, NULL);
    if (rc == LLZ_OK) rc = llzs_memset(f->d_hist[1], 0, hist_bytes, NULL);
    if (rc == LLZ_OK) rc = llzs_memset(f->d_zero, 0, hist_bytes, NULL);
    if (rc == LLZ_OK && algo == LLZ_FIR_ALGO_OVERLAP_SAVE) rc = firm_build_ols_tables(f, taps);
    if (rc == LLZ_OK && algo == LLZ_FIR_ALGO_OVERLAP_SAVE_2048) rc = firm_build_ols2k_tables(f, taps);
    if (rc == LLZ_OK && algo == LLZ_FIR_ALGO_OVERLAP_SAVE_4096)
        rc = firm_build_ols4k_tables(f, taps);
    if (rc == LLZ_OK && algo == LLZ_FIR_ALGO_OVERLAP_SAVE_8192) rc = firm_build_ols8k_tables(f, taps);
    if (rc == LLZ_OK) rc = llzs_sync(NULL);
    free(padded);
    if (rc != LLZ_OK) {
        firm_destroy(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

unsigned long llz_fir_filter_mc_init_f64taps(int channels, int frame_len, const double *taps, int flt_len, int algo)
{
    if (!taps || flt_len < 1) {
        llzs_set_error("llz_fir_filter_mc_init_f64taps: no taps");
        return LLZ_BAD_HANDLE;
    }
    float *t = (float *)malloc(sizeof(float) * (size_t)flt_len);
    if (!t) return LLZ_BAD_HANDLE;
    for (int i = 0; i < flt_len; i++) t[i] = (float)taps[i];
    unsigned long h = llz_fir_filter_mc_init(channels, frame_len, t, flt_len, algo);
    free(t);
    return h;
}

static unsigned long firm_design_init(int kind, int channels, int frame_len, int flt_len, double fc1, double fc2,
                                      win_t win)
{
    double *h = NULL;
    const int n = llz_host_design(kind, &h, flt_len, fc1, fc2, win);
    if (n < 1) {
        llzs_set_error("llz_fir_filter_mc_*_init: tap design failed");
        return LLZ_BAD_HANDLE;
    }
    unsigned long handle = llz_fir_filter_mc_init_f64taps(channels, frame_len, h, n, LLZ_FIR_ALGO_AUTO);
    free(h);
    return handle;
}

unsigned long llz_fir_filter_mc_lpf_init(int channels, int frame_len, int flt_len, double fc, win_t win_type)
{
    return firm_design_init(LLZ_KIND_LPF, channels, frame_len, flt_len, fc, 0.0, win_type);
}

unsigned long llz_fir_filter_mc_hpf_init(int channels, int frame_len, int flt_len, double fc, win_t win_type)
{
    return firm_design_init(LLZ_KIND_HPF, channels, frame_len, flt_len, fc, 0.0, win_type);
}

unsigned long llz_fir_filter_mc_bandpass_init(int channels, int frame_len, int flt_len, double fc1, double fc2,
                                              win_t win_type)
{
    return firm_design_init(LLZ_KIND_BPF, channels, frame_len, flt_len, fc1, fc2, win_type);
}

unsigned long llz_fir_filter_mc_bandstop_init(int channels, int frame_len, int flt_len, double fc1, double fc2,
                                              win_t win_type)
{
    return firm_design_init(LLZ_KIND_BSF, channels, frame_len, flt_len, fc1, fc2, win_type);
}

void llz_fir_filter_mc_uninit(unsigned long handle)
{
    if (LLZ_HANDLE_OK(handle, firm_t, LLZ_TAG_FIRM)) {
        firm_t *f = (firm_t *)handle;
        const int prev = llzs_device_enter(f->device);
        llzs_sync(f->stream);
        firm_destroy(f);
        llzs_device_leave(prev);
    }
}

int llz_fir_filter_mc_flt_len(unsigned long handle)
{
    return LLZ_HANDLE_OK(handle, firm_t, LLZ_TAG_FIRM) ? ((firm_t *)handle)->flt_len : LLZ_ERR_ARG;
}

int llz_fir_filter_mc_algo(unsigned long handle)
{
    return LLZ_HANDLE_OK(handle, firm_t, LLZ_TAG_FIRM) ? ((firm_t *)handle)->algo : LLZ_ERR_ARG;
}

int llz_fir_filter_mc_set_stream(unsigned long handle, void *stream)
{
    if (!LLZ_HANDLE_OK(handle, firm_t, LLZ_TAG_FIRM)) return LLZ_ERR_ARG;
    ((firm_t *)handle)->stream = stream;
    return LLZ_OK;
}

/* device-side body shared by process and flush */
static int firm_launch(firm_t *f, const float *d_in, float *d_out, int n, long pitch_in, long pitch_out, int algo)
{
    const float *hist = f->flt_len > 1 ? f->d_hist[f->cur] : NULL;
    int rc;
    if (algo == LLZ_FIR_ALGO_OVERLAP_SAVE)
        rc = llzs_fir_ols_f32(d_in, d_out, hist, f->d_hfreq, f->d_twid, f->channels, n, pitch_in, pitch_out,
                              f->flt_len, f->stream);
    else if (algo == LLZ_FIR_ALGO_OVERLAP_SAVE_2048)
        rc = llzs_fir_ols2k_f32(d_in, d_out, hist, f->d_hfreq, f->d_twid, f->d_tw2k, f->channels, n, pitch_in, pitch_out,
                                f->flt_len, f->stream);
    else if (algo == LLZ_FIR_ALGO_OVERLAP_SAVE_4096)
        rc = llzs_fir_ols4k_f32(d_in, d_out, hist, f->d_hfreq, f->d_twid, f->d_tw2k, f->d_tw4k, f->channels, n, pitch_in,
                                pitch_out, f->flt_len, f->stream);
    else if (algo == LLZ_FIR_ALGO_OVERLAP_SAVE_8192)
        rc = llzs_fir_ols8k_f32(d_in, d_out, hist, f->d_hfreq, f->d_twid, f->d_tw4k, f->channels, n, pitch_in, pitch_out,
                                f->flt_len, f->stream);
    else if (algo == LLZ_FIR_ALGO_TIME_MFMA)
        rc = llzs_fir_mfma_f32(d_in, d_out, hist, f->d_taps, f->channels, n, n, pitch_in, pitch_out, f->flt_len, 1,
                               1.0f, f->stream);
    else
        rc = llzs_fir_td_f32(d_in, d_out, hist, f->d_taps, f->channels, n, pitch_in, pitch_out, f->flt_len,
                             f->stream);
    if (rc == LLZ_OK && f->flt_len > 1) {
        rc = llzs_fir_tail_f32(d_in, f->d_hist[f->cur], f->d_hist[f->cur ^ 1], f->channels, n, pitch_in,
                               f->flt_len, f->stream);
        if (rc == LLZ_OK) f->cur ^= 1;
    }
    return rc;
}

static int firm_process(firm_t *f, const float *in, float *out, int frame_len);
static int firm_flush(firm_t *f, float *out);

int llz_fir_filter_mc(unsigned long handle, const float *in, float *out, int frame_len)
{
    if (!LLZ_HANDLE_OK(handle, firm_t, LLZ_TAG_FIRM) || !in || !out) {
        llzs_set_error("llz_fir_filter_mc: bad handle or NULL buffer");
        return LLZ_ERR_ARG;
    }
    firm_t *f = (firm_t *)handle;
    const int prev = llzs_device_enter(f->device);       /* the handle's device, whatever the caller has current */
    const int rc = firm_process(f, in, out, frame_len);
    llzs_device_leave(prev);
    return rc;
}

static int firm_process(firm_t *f, const float *in, float *out, int frame_len)
{
    if (frame_len != f->frame_len) {
        llzs_set_error("llz_fir_filter_mc: frame_len %d != init frame_len %d", frame_len, f->frame_len);
        return LLZ_ERR_ARG;
    }
    if (in == out) {
        llzs_set_error("llz_fir_filter_mc: in-place filtering is not supported");
        return LLZ_ERR_ARG;
    }
    const size_t bytes = sizeof(float) * (size_t)f->channels * (size_t)frame_len;
    const int in_dev = llzs_is_device_ptr(in), out_dev = llzs_is_device_ptr(out);
    if (in_dev < 0 || out_dev < 0) return LLZ_ERR_ARG;            /* a buffer of another GPU: refused, message set */
    const float *d_in = in;
    float *d_out = out;
    int rc = LLZ_OK;
    if (!in_dev) {
        d_in = (const float *)llz_stage_reserve(&f->st_in, bytes);
        if (!d_in) return LLZ_ERR_NOMEM;
        rc = llzs_h2d((void *)d_in, in, bytes, f->stream);
    }
    if (rc == LLZ_OK && !out_dev) {
        d_out = (float *)llz_stage_reserve(&f->st_out, bytes);
        if (!d_out) return LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK) rc = firm_launch(f, d_in, d_out, frame_len, frame_len, frame_len, f->algo);
    if (rc == LLZ_OK && !out_dev) rc = llzs_d2h(out, d_out, bytes, f->stream);
    return rc == LLZ_OK ? frame_len : rc;
}

int llz_fir_filter_mc_flush(unsigned long handle, float *out)
{
    if (!LLZ_HANDLE_OK(handle, firm_t, LLZ_TAG_FIRM) || !out) {
        llzs_set_error("llz_fir_filter_mc_flush: bad handle or NULL buffer");
        return LLZ_ERR_ARG;
    }
    firm_t *f = (firm_t *)handle;
    const int prev = llzs_device_enter(f->device);
    const int rc = firm_flush(f, out);
    llzs_device_leave(prev);
    return rc;
}

static int firm_flush(firm_t *f, float *out)
{
    const int keep = f->flt_len - 1;
    if (keep == 0) return 0;
    const size_t bytes = sizeof(float) * (size_t)f->channels * (size_t)keep;
    const int out_dev = llzs_is_device_ptr(out);
    if (out_dev < 0) return LLZ_ERR_ARG;
    float *d_out = out;
    if (!out_dev) {
        d_out = (float *)llz_stage_reserve(&f->st_out, bytes);
        if (!d_out) return LLZ_ERR_NOMEM;
    }
    /* flt_len-1 zeros per channel through the time-domain kernel (tiny; same arithmetic as the frames) */
    int rc = firm_launch(f, f->d_zero, d_out, keep, keep, keep, LLZ_FIR_ALGO_TIME);
    if (rc == LLZ_OK && !out_dev) rc = llzs_d2h(out, d_out, bytes, f->stream);
    return rc == LLZ_OK ? keep : rc;
}
