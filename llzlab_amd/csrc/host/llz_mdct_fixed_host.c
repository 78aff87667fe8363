/*
 * llz_mdct_fixed_host.c -- handle layer of the fixed-point MDCT (fixed half): the reference's
 * symbols (reference libllzfilter/llz_mdct_fixed.h:24-28, llz_mdct_fixed.c:116-526) and their batch form.
 *
 * The host builds the Q15 tables (libm + the reference's rounding macro, so the table entries are the reference's to the
 * last bit), uploads them once, and moves frames; all arithmetic on the data runs on the device (kernels/mdct_q15.hip
 * around the bit-exact Q15 transform of kernels/fft.hip).  One handle serves one frame per call through the reference's
 * symbols and `count` frames per call through llz_mdct_fixed_batch / llz_imdct_fixed_batch.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../../include/llz_mdct_fixed.h"
#include "llz_host.h"

#define LLZ_TAG_MDCX 0x4c5a4d58

enum { STEP_FWD_PRE = 0, STEP_FWD_POST = 1, STEP_INV_PRE = 2, STEP_INV_POST = 3 };

typedef struct {
    int tag, form, length, device;
    int fft_size;                       /* points of the transform inside the two FFT forms */
    int cof;                            /* Q15 1 / sqrt(length): the N/4-point form's inverse scale */
    short *d_sum_fwd, *d_sum_inv;       /* MDCT_FIXED_ORIGIN: [N/2][N] and [N][N/2] Q15 cosine matrices */
    short *d_step[4];                   /* FFT forms: (cos, sin) Q15 pair tables of the four twiddle steps */
    short *d_fft_cs;                    /* fft_size cos then fft_size sin, Q15 (llz_fft_fixed.c:243-247) */
    llz_stage_t time, bins, work;       /* device frames: count x N samples, count x N/2 coefficients, count x points */
    void *stream;
} mdcx_t;

/* LLZ_FIX15 of the reference (llz_fft_fixed.h:42-66): v * 2^15 rounded half away from zero, saturated to int32, clipped to
 * +-32767 */
static short q15_of(double v)
{
    const double t = v * (double)(1 << 15);
    const double r = (t > 0) ? floor(t + 0.5) : ceil(t - 0.5);
    int q = r > 2147483647.0 ? 2147483647 : r < -2147483648.0 ? (-2147483647 - 1) : (int)r;
    if (q > 32767) q = 32767;
    if (q < -32767) q = -32767;
    return (short)q;
}

static void mdcx_destroy(mdcx_t *f)
{
    if (!f) return;
    llzs_free(f->d_sum_fwd); llzs_free(f->d_sum_inv);
    for (int t = 0; t < 4; t++) {
        int shared = 0;
        for (int u = 0; u < t; u++) shared |= (f->d_step[u] == f->d_step[t]);
        if (!shared) llzs_free(f->d_step[t]);
    }
    llzs_free(f->d_fft_cs);
    llz_stage_release(&f->time); llz_stage_release(&f->bins); llz_stage_release(&f->work);
    f->tag = 0;
    free(f);
}

/* angle of entry k of a step's table: the reference's expressions (llz_mdct_fixed.c:335-366, :381-386) in its evaluation
 * order -- the Q15 rounding of cos/sin of a differently rounded angle could differ in the last bit */
static double step_angle(int form, int step, int k, int length)
{
    if (form == MDCT_FIXED_FFT4) return -2 * M_PI * (k + 0.125) / length;
    const double n0 = ((double)length / 2 + 1) / 2;
    switch (step) {
    case STEP_FWD_PRE:  return -(M_PI * k) / length;
    case STEP_FWD_POST: return -2 * M_PI * n0 * (k + 0.5) / length;
    case STEP_INV_PRE:  return (2 * M_PI * k * n0) / length;
    default:            return M_PI * (k + n0) / length;
    }
}

static short *upload_q15(const short *host, size_t count)
{
    short *dev = (short *)llzs_malloc(sizeof(short) * count);
    if (dev && llzs_h2d(dev, host, sizeof(short) * count, NULL) != LLZ_OK) { llzs_free(dev); dev = NULL; }
    return dev;
}

static short *step_table_upload(int form, int step, int count, int length)
{
    short *host = (short *)malloc(sizeof(short) * 2 * (size_t)count);
    if (!host) return NULL;
    for (int k = 0; k < count; k++) {
        const double ang = step_angle(form, step, k, length);
        host[2 * k] = q15_of(cos(ang));
        host[2 * k + 1] = q15_of(sin(ang));
    }
    short *dev = upload_q15(host, 2 * (size_t)count);
    free(host);
    return dev;
}

static int mdcx_build_sums(mdcx_t *f)
{
    const int N = f->length, K = N >> 1;
    const size_t cnt = (size_t)K * N;
    short *fwd = (short *)malloc(sizeof(short) * cnt), *inv = (short *)malloc(sizeof(short) * cnt);
    int rc = (fwd && inv) ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK) {
        for (int k = 0; k < K; k++)
            for (int n = 0; n < N; n++) {
                const double ang = (M_PI / (2 * N)) * (2 * n + 1 + K) * (2 * k + 1);    /* llz_mdct_fixed.c:313-322 */
                fwd[(size_t)k * N + n] = inv[(size_t)n * K + k] = q15_of(cos(ang));
            }
        f->d_sum_fwd = upload_q15(fwd, cnt);
        f->d_sum_inv = upload_q15(inv, cnt);
        if (!f->d_sum_fwd || !f->d_sum_inv) rc = LLZ_ERR_NOMEM;
    }
    free(fwd); free(inv);
    return rc;
}

static int mdcx_build_fft_form(mdcx_t *f)
{
    const int N = f->length;
    f->fft_size = f->form == MDCT_FIXED_FFT ? N : N >> 2;
    if (f->form == MDCT_FIXED_FFT) {
        f->d_step[STEP_FWD_PRE] = step_table_upload(MDCT_FIXED_FFT, STEP_FWD_PRE, N, N);
        f->d_step[STEP_FWD_POST] = step_table_upload(MDCT_FIXED_FFT, STEP_FWD_POST, N >> 1, N);
        f->d_step[STEP_INV_PRE] = step_table_upload(MDCT_FIXED_FFT, STEP_INV_PRE, N, N);
        f->d_step[STEP_INV_POST] = step_table_upload(MDCT_FIXED_FFT, STEP_INV_POST, N, N);
    } else {
        short *t = step_table_upload(MDCT_FIXED_FFT4, 0, N >> 2, N);      /* one table serves all four steps */
        for (int s = 0; s < 4; s++) f->d_step[s] = t;
        f->cof = q15_of(1. / sqrt(N));                                    /* llz_mdct_fixed.c:373 */
    }
    for (int s = 0; s < 4; s++)
        if (!f->d_step[s]) return LLZ_ERR_NOMEM;
    const int F = f->fft_size;
    short *cs = (short *)malloc(sizeof(short) * 2 * (size_t)F);
    if (!cs) return LLZ_ERR_NOMEM;
    for (int i = 0; i < F; i++) {
        const double ang = (2 * M_PI * i) / F;                            /* llz_fft_fixed.c:243-247 */
        cs[i] = q15_of(cos(ang));
        cs[F + i] = q15_of(sin(ang));
    }
    f->d_fft_cs = upload_q15(cs, 2 * (size_t)F);
    free(cs);
    return f->d_fft_cs ? LLZ_OK : LLZ_ERR_NOMEM;
}

unsigned long llz_mdct_fixed_init(int type, int size)
{
    if (size < 4 || (type != MDCT_FIXED_ORIGIN && type != MDCT_FIXED_FFT && type != MDCT_FIXED_FFT4)) {
        llzs_set_error("llz_mdct_fixed_init: type %d len %d", type, size);
        return LLZ_BAD_HANDLE;
    }
    int log2len = (int)(log(size) / log(2));                        /* next power of two, as llz_mdct_fixed.c:296-300 */
    if ((1 << log2len) < size) log2len += 1;
    const int length = 1 << log2len;
    const int limit = type == MDCT_FIXED_ORIGIN ? 2048 : (type == MDCT_FIXED_FFT ? 4096 : 16384);
    if (length > limit || (type == MDCT_FIXED_FFT4 && length < 8)) {
        llzs_set_error("llz_mdct_fixed_init: length %d out of range for type %d (at most %d)", length, type, limit);
        return LLZ_BAD_HANDLE;
    }
    mdcx_t *f = (mdcx_t *)calloc(1, sizeof(*f));
    if (!f) return LLZ_BAD_HANDLE;
    f->tag = LLZ_TAG_MDCX; f->form = type; f->length = length; f->device = llzs_device_get();
    int rc = type == MDCT_FIXED_ORIGIN ? mdcx_build_sums(f) : mdcx_build_fft_form(f);
    if (rc == LLZ_OK && f->device < 0) rc = LLZ_ERR_DEVICE;
    if (rc != LLZ_OK) {
        mdcx_destroy(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

void llz_mdct_fixed_uninit(unsigned long handle)
{
    if (!LLZ_HANDLE_OK(handle, mdcx_t, LLZ_TAG_MDCX)) return;
    mdcx_t *f = (mdcx_t *)handle;
    const int prev = llzs_device_enter(f->device);
    llzs_sync(f->stream);
    mdcx_destroy(f);
    llzs_device_leave(prev);
}

int llz_mdct_fixed_set_stream(unsigned long handle, void *stream)
{
    if (!LLZ_HANDLE_OK(handle, mdcx_t, LLZ_TAG_MDCX)) return LLZ_ERR_ARG;
    ((mdcx_t *)handle)->stream = stream;
    return LLZ_OK;
}

int llz_mdct_fixed_len(unsigned long handle)
{
    return LLZ_HANDLE_OK(handle, mdcx_t, LLZ_TAG_MDCX) ? ((mdcx_t *)handle)->length : LLZ_ERR_ARG;
}

/* `count` frames in one direction, device buffers: d_src -> d_dst, through the handle's work buffer */
static int mdcx_on_device(mdcx_t *f, const int *d_src, int *d_dst, int count, int inverse)
{
    const int N = f->length, K = N >> 1;
    if (f->form == MDCT_FIXED_ORIGIN)                                /* llz_mdct_fixed.c:116-152 */
        return inverse ? llzs_mdctq_sums(f->d_sum_inv, d_src, d_dst, count, N, K, N, f->stream)
                       : llzs_mdctq_sums(f->d_sum_fwd, d_src, d_dst, count, K, N, 0, f->stream);
    const int quarter = f->form == MDCT_FIXED_FFT4;
    if (quarter && N >= 8 && N <= 16384 && llzs_tune(LLZS_TUNE_MDCTQ_STEPS) < 1)   /* one launch, the frame in and out of HBM once */
        return llzs_mdct4_q15(d_src, d_dst, count, N, f->d_step[inverse ? STEP_INV_PRE : STEP_FWD_PRE],
                              f->d_step[inverse ? STEP_INV_POST : STEP_FWD_POST], f->d_fft_cs, inverse, f->cof, f->stream);
    if (!quarter && N >= 4 && N <= 4096 && llzs_tune(LLZS_TUNE_MDCTQ_STEPS) < 1)   /* the N-point form likewise */
        return llzs_mdct1_q15(d_src, d_dst, count, N, f->d_step[inverse ? STEP_INV_PRE : STEP_FWD_PRE],
                              f->d_step[inverse ? STEP_INV_POST : STEP_FWD_POST], f->d_fft_cs, inverse, f->stream);
    int *d_work = (int *)llz_stage_reserve(&f->work, sizeof(int) * 2 * (size_t)f->fft_size * (size_t)count);
    if (!d_work) return LLZ_ERR_NOMEM;
    /* the N-point form inverts with the inverse transform (llz_mdct_fixed.c:190); the N/4-point form runs the FORWARD
     * transform in both directions (llz_mdct_fixed.c:223, :258) */
    const int fft_inverse = inverse && !quarter;
    int rc = llzs_mdctq_step(quarter, 0, d_src, d_work, f->d_step[inverse ? STEP_INV_PRE : STEP_FWD_PRE], count, N,
                             inverse, f->cof, f->stream);
    if (rc == LLZ_OK) rc = llzs_fft_fixed(d_work, count, f->fft_size, f->d_fft_cs, fft_inverse, f->stream);
    if (rc == LLZ_OK)
        rc = llzs_mdctq_step(quarter, 1, d_work, d_dst, f->d_step[inverse ? STEP_INV_POST : STEP_FWD_POST], count, N,
                             inverse, f->cof, f->stream);
    return rc;
}

static int mdcx_run(unsigned long handle, const int *in, int *out, int count, int inverse, const char *who)
{
    if (!LLZ_HANDLE_OK(handle, mdcx_t, LLZ_TAG_MDCX) || !in || !out || in == out || count < 1) {
        llzs_set_error("%s: bad handle or arguments (at least one frame, out of place)", who);
        return LLZ_ERR_ARG;
    }
    mdcx_t *f = (mdcx_t *)handle;
    const int prev = llzs_device_enter(f->device);
    const size_t full = sizeof(int) * (size_t)count * (size_t)f->length, half = full / 2;
    const size_t ib = inverse ? half : full, ob = inverse ? full : half;
    const int in_dev = llzs_is_device_ptr(in), out_dev = llzs_is_device_ptr(out);
    const int *d_in = in;
    int *d_out = out;
    int rc = (in_dev < 0 || out_dev < 0) ? LLZ_ERR_ARG : LLZ_OK;
    if (rc == LLZ_OK && !in_dev) {
        d_in = (const int *)llz_stage_reserve(inverse ? &f->bins : &f->time, ib);
        rc = d_in ? llzs_h2d((void *)d_in, in, ib, f->stream) : LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK && !out_dev) {
        d_out = (int *)llz_stage_reserve(inverse ? &f->time : &f->bins, ob);
        if (!d_out) rc = LLZ_ERR_NOMEM;
    }
    /* (a launch of the step kernels takes at most 65535 frames: the frame index is the grid's second dimension; the one-launch
     * N/4-point form takes any number) */
    const int whole = f->form != MDCT_FIXED_ORIGIN && llzs_tune(LLZS_TUNE_MDCTQ_STEPS) < 1;
    const int most = whole ? count : 65535;
    for (int done = 0; rc == LLZ_OK && done < count; done += most) {
        const int part = count - done < most ? count - done : most;
        const size_t n_in = inverse ? (size_t)f->length / 2 : (size_t)f->length, n_out = inverse ? (size_t)f->length : (size_t)f->length / 2;
        rc = mdcx_on_device(f, d_in + (size_t)done * n_in, d_out + (size_t)done * n_out, part, inverse);
    }
    if (rc == LLZ_OK && !out_dev) rc = llzs_d2h(out, d_out, ob, f->stream);
    llzs_device_leave(prev);
    return rc == LLZ_OK ? count : rc;
}

int llz_mdct_fixed_batch(unsigned long handle, const int *x, int *X, int count)
{
    return mdcx_run(handle, x, X, count, 0, "llz_mdct_fixed_batch");
}

int llz_imdct_fixed_batch(unsigned long handle, const int *X, int *x, int count)
{
    return mdcx_run(handle, X, x, count, 1, "llz_imdct_fixed_batch");
}

/* the reference's symbols: one frame on host buffers (void in the reference ABI: errors are left in llz_hip_last_error) */
void llz_mdct_fixed(unsigned long handle, int *x, int *X)
{
    (void)mdcx_run(handle, x, X, 1, 0, "llz_mdct_fixed");
}

void llz_imdct_fixed(unsigned long handle, int *X, int *x)
{
    (void)mdcx_run(handle, X, x, 1, 1, "llz_imdct_fixed");
}
