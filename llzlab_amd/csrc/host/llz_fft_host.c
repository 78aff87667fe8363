/*
 * llz_fft_host.c -- handle layer of the FFT path: reference symbols llz_fft_* (reference libllzfilter/llz_fft.c:142-249)
 * and llz_fft_fixed_* (llz_fft_fixed.c:152-268) plus the batched float32 / int32 entry points.  Twiddle tables are
 * built here on the host exactly as the reference builds them and uploaded once per handle.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../../include/llz_fft.h"
#include "../../../include/llz_fft_fixed.h"
#include "llz_host.h"

static int is_pow2_in_range(int size, int lo, int hi)
{
    return size >= lo && size <= hi && (size & (size - 1)) == 0;
}

/* ---- double, single transform ---- */

typedef struct {
    int tag, size, device;
    double *d_cs;       /* size cos then size sin */
    double *d_data;     /* 2*size doubles */
} fft1_t;

double *llz_host_fft_table_f64(int size)
{
    double *cs = (double *)malloc(sizeof(double) * 2 * (size_t)size);
    double *dev = (double *)llzs_malloc(sizeof(double) * 2 * (size_t)size);
    if (cs && dev) {
        for (int i = 0; i < size; i++) {
            const double ang = (double)(2 * M_PI * i) / size;    /* llz_fft.c:223-227 */
            cs[i] = cos(ang);
            cs[size + i] = sin(ang);
        }
        if (llzs_h2d(dev, cs, sizeof(double) * 2 * (size_t)size, NULL) != LLZ_OK) { llzs_free(dev); dev = NULL; }
    } else {
        llzs_free(dev);
        dev = NULL;
    }
    free(cs);
    return dev;
}

unsigned long llz_fft_init(int size)
{
    if (!is_pow2_in_range(size, 2, 4096)) {
        llzs_set_error("llz_fft_init: size %d must be a power of two in 2..4096", size);
        return LLZ_BAD_HANDLE;
    }
    fft1_t *f = (fft1_t *)calloc(1, sizeof(*f));
    if (!f) return LLZ_BAD_HANDLE;
    f->tag = LLZ_TAG_FFT1; f->size = size; f->device = llzs_device_get();
    f->d_cs = llz_host_fft_table_f64(size);
    f->d_data = (double *)llzs_malloc(sizeof(double) * 2 * (size_t)size);
    if (!f->d_cs || !f->d_data || f->device < 0) {
        llzs_free(f->d_cs); llzs_free(f->d_data); free(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

void llz_fft_uninit(unsigned long handle)
{
    if (!LLZ_HANDLE_OK(handle, fft1_t, LLZ_TAG_FFT1)) return;
    fft1_t *f = (fft1_t *)handle;
    const int prev = llzs_device_enter(f->device);
    llzs_sync(NULL);
    llzs_free(f->d_cs); llzs_free(f->d_data);
    llzs_device_leave(prev);
    f->tag = 0;
    free(f);
}

static void fft1_run(unsigned long handle, double *data, int inverse)
{
    if (!LLZ_HANDLE_OK(handle, fft1_t, LLZ_TAG_FFT1) || !data) {
        llzs_set_error("llz_fft/llz_ifft: bad handle or NULL data");
        return;                                                   /* void in the reference ABI */
    }
    fft1_t *f = (fft1_t *)handle;
    const size_t bytes = sizeof(double) * 2 * (size_t)f->size;
    const int prev = llzs_device_enter(f->device);
    int rc = llzs_h2d(f->d_data, data, bytes, NULL);
    if (rc == LLZ_OK) rc = llzs_fft_f64(f->d_data, f->size, f->d_cs, inverse, NULL);
    if (rc == LLZ_OK) (void)llzs_d2h(data, f->d_data, bytes, NULL);
    llzs_device_leave(prev);
}

void llz_fft(unsigned long handle, double *data)  { fft1_run(handle, data, 0); }
void llz_ifft(unsigned long handle, double *data) { fft1_run(handle, data, 1); }

/* ---- float32 batch ---- */

typedef struct {
    int tag, size, device;
    float *d_cs;
    void *stream;
    llz_stage_t st;
} fftb_t;

unsigned long llz_fft_batch_init(int size)
{
    if (!is_pow2_in_range(size, 8, 4096)) {
        llzs_set_error("llz_fft_batch_init: size %d must be a power of two in 8..4096", size);
        return LLZ_BAD_HANDLE;
    }
    fftb_t *f = (fftb_t *)calloc(1, sizeof(*f));
    float *cs = (float *)malloc(sizeof(float) * 2 * (size_t)size);
    if (!f || !cs) { free(f); free(cs); return LLZ_BAD_HANDLE; }
    f->tag = LLZ_TAG_FFTB; f->size = size; f->device = llzs_device_get();
    for (int i = 0; i < size; i++) {
        const double ang = (double)(2 * M_PI * i) / size;
        cs[i] = (float)cos(ang);
        cs[size + i] = (float)sin(ang);
    }
    f->d_cs = (float *)llzs_malloc(sizeof(float) * 2 * (size_t)size);
    const int ok = f->d_cs && llzs_h2d(f->d_cs, cs, sizeof(float) * 2 * (size_t)size, NULL) == LLZ_OK;
    free(cs);
    if (!ok) { llzs_free(f->d_cs); free(f); return LLZ_BAD_HANDLE; }
    return (unsigned long)f;
}

void llz_fft_batch_uninit(unsigned long handle)
{
    if (!LLZ_HANDLE_OK(handle, fftb_t, LLZ_TAG_FFTB)) return;
    fftb_t *f = (fftb_t *)handle;
    const int prev = llzs_device_enter(f->device);
    llzs_sync(f->stream);
    llzs_free(f->d_cs);
    llz_stage_release(&f->st);
    llzs_device_leave(prev);
    f->tag = 0;
    free(f);
}

int llz_fft_batch_set_stream(unsigned long handle, void *stream)
{
    if (!LLZ_HANDLE_OK(handle, fftb_t, LLZ_TAG_FFTB)) return LLZ_ERR_ARG;
    ((fftb_t *)handle)->stream = stream;
    return LLZ_OK;
}

static int fftb_run(unsigned long handle, float *data, int count, int inverse)
{
    if (!LLZ_HANDLE_OK(handle, fftb_t, LLZ_TAG_FFTB) || !data || count < 1) {
        llzs_set_error("llz_fft_batch: bad handle, NULL data or count %d", count);
        return LLZ_ERR_ARG;
    }
    fftb_t *f = (fftb_t *)handle;
    const size_t bytes = sizeof(float) * 2 * (size_t)f->size * (size_t)count;
    const int prev = llzs_device_enter(f->device);
    const int on_dev = llzs_is_device_ptr(data);
    int rc = on_dev < 0 ? LLZ_ERR_ARG : LLZ_OK;
    if (rc == LLZ_OK && on_dev) {
        rc = llzs_fft_f32(data, count, f->size, f->d_cs, inverse, f->stream);
    } else if (rc == LLZ_OK) {
        float *d = (float *)llz_stage_reserve(&f->st, bytes);
        rc = d ? llzs_h2d(d, data, bytes, f->stream) : LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) rc = llzs_fft_f32(d, count, f->size, f->d_cs, inverse, f->stream);
        if (rc == LLZ_OK) rc = llzs_d2h(data, d, bytes, f->stream);
    }
    llzs_device_leave(prev);
    return rc;
}

int llz_fft_batch(unsigned long handle, float *data, int count)  { return fftb_run(handle, data, count, 0); }
int llz_ifft_batch(unsigned long handle, float *data, int count) { return fftb_run(handle, data, count, 1); }

/* ---- fixed point ---- */

typedef struct {
    int tag, size, device;
    short *d_cs;
    void *stream;
    llz_stage_t st;
} fftx_t;

/* llz_fft_fixed.h:42-66: round half away from zero of v * 2^15, saturate to int32, clip to +-32767 */
static short q15_round(double v)
{
    const double t = v * (double)(1 << 15);
    const double r = (t > 0) ? floor(t + 0.5) : ceil(t - 0.5);
    int q = r > 2147483647.0 ? 2147483647 : r < -2147483648.0 ? (-2147483647 - 1) : (int)r;
    if (q > 32767) q = 32767;
    if (q < -32767) q = -32767;
    return (short)q;
}

unsigned long llz_fft_fixed_init(int size)
{
    if (!is_pow2_in_range(size, 2, 4096)) {
        llzs_set_error("llz_fft_fixed_init: size %d must be a power of two in 2..4096", size);
        return LLZ_BAD_HANDLE;
    }
    fftx_t *f = (fftx_t *)calloc(1, sizeof(*f));
    short *cs = (short *)malloc(sizeof(short) * 2 * (size_t)size);
    if (!f || !cs) { free(f); free(cs); return LLZ_BAD_HANDLE; }
    f->tag = LLZ_TAG_FFTX; f->size = size; f->device = llzs_device_get();
    for (int i = 0; i < size; i++) {
        const double ang = (2 * M_PI * i) / size;                 /* llz_fft_fixed.c:243-247 */
        cs[i] = q15_round(cos(ang));
        cs[size + i] = q15_round(sin(ang));
    }
    f->d_cs = (short *)llzs_malloc(sizeof(short) * 2 * (size_t)size);
    const int ok = f->d_cs && llzs_h2d(f->d_cs, cs, sizeof(short) * 2 * (size_t)size, NULL) == LLZ_OK;
    free(cs);
    if (!ok) { llzs_free(f->d_cs); free(f); return LLZ_BAD_HANDLE; }
    return (unsigned long)f;
}

void llz_fft_fixed_uninit(unsigned long handle)
{
    if (!LLZ_HANDLE_OK(handle, fftx_t, LLZ_TAG_FFTX)) return;
    fftx_t *f = (fftx_t *)handle;
    const int prev = llzs_device_enter(f->device);
    llzs_sync(f->stream);
    llzs_free(f->d_cs);
    llz_stage_release(&f->st);
    llzs_device_leave(prev);
    f->tag = 0;
    free(f);
}

int llz_fft_fixed_set_stream(unsigned long handle, void *stream)
{
    if (!LLZ_HANDLE_OK(handle, fftx_t, LLZ_TAG_FFTX)) return LLZ_ERR_ARG;
    ((fftx_t *)handle)->stream = stream;
    return LLZ_OK;
}

static int fftx_run(unsigned long handle, int *data, int count, int inverse)
{
    if (!LLZ_HANDLE_OK(handle, fftx_t, LLZ_TAG_FFTX) || !data || count < 1) {
        llzs_set_error("llz_fft_fixed: bad handle, NULL data or count %d", count);
        return LLZ_ERR_ARG;
    }
    fftx_t *f = (fftx_t *)handle;
    const size_t bytes = sizeof(int) * 2 * (size_t)f->size * (size_t)count;
    const int prev = llzs_device_enter(f->device);
    const int on_dev = llzs_is_device_ptr(data);
    int rc = on_dev < 0 ? LLZ_ERR_ARG : LLZ_OK;
    if (rc == LLZ_OK && on_dev) {
        rc = llzs_fft_fixed(data, count, f->size, f->d_cs, inverse, f->stream);
    } else if (rc == LLZ_OK) {
        int *d = (int *)llz_stage_reserve(&f->st, bytes);
        rc = d ? llzs_h2d(d, data, bytes, f->stream) : LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) rc = llzs_fft_fixed(d, count, f->size, f->d_cs, inverse, f->stream);
        if (rc == LLZ_OK) rc = llzs_d2h(data, d, bytes, f->stream);
    }
    llzs_device_leave(prev);
    return rc;
}

void llz_fft_fixed(unsigned long handle, int *data)  { (void)fftx_run(handle, data, 1, 0); }
void llz_ifft_fixed(unsigned long handle, int *data) { (void)fftx_run(handle, data, 1, 1); }
int llz_fft_fixed_batch(unsigned long handle, int *data, int count)  { return fftx_run(handle, data, count, 0); }
int llz_ifft_fixed_batch(unsigned long handle, int *data, int count) { return fftx_run(handle, data, count, 1); }
