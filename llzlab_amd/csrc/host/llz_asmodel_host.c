/*
 * llz_asmodel_host.c -- handle layer of the windowed-FFT analysis / synthesis frames: the
 * reference's symbols (reference libllzfilter/llz_asmodel.c:109-310) and the float32 batch extension.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../../include/llz_asmodel.h"
#include "../../../include/llz_mdct.h"
#include "llz_host.h"

#define LLZ_TAG_ASM1 0x4c5a5331
#define LLZ_TAG_ASMM 0x4c5a534d

static int asm_shape(int overlap_hint, int frame_len, int *fft_len, double *magic)
{
    /* llz_asmodel.c:114-123 */
    if (overlap_hint == LLZ_OVERLAP_HIGH) { *fft_len = frame_len << 2; *magic = 0.812; }
    else if (overlap_hint == LLZ_OVERLAP_LOW) { *fft_len = frame_len << 1; *magic = 1; }
    else return 0;
    return frame_len >= 1 && *fft_len >= 2 && (*fft_len & (*fft_len - 1)) == 0;
}

static int asm_window(double *w, int n, win_t win_type)
{
    switch (win_type) {                                            /* llz_asmodel.c:133-143 */
    case HAMMING: llz_hamming(w, n); return 1;
    case BLACKMAN: llz_blackman(w, n); return 1;
    case KAISER: llz_kaiser(w, n); return 1;
    default: return 0;
    }
}

/* ---- Part 1: the reference's symbols, one channel, one frame per call ----
 * A handle keeps its running frame buffer, the window and the scratch spectrum in DEVICE memory.  A call copies the new
 * samples (or the two spectrum planes) in, runs framing kernel -> exact-order transform -> framing kernel on the device
 * (kernels/frames_f64.hip, kernels/fft.hip), and copies the result out; nothing is computed on the host. */

typedef struct {
    int tag, device, hop, size, bins;       /* hop = frame_len, size = fft_len, bins = size/2 + 1 */
    double out_scale;                       /* the reference's empirical 0.812 at 3/4 overlap, 1 at 1/2 */
    double *d_window, *d_fft_cs;
    double *d_run[2];                       /* analysis: the last `size` input samples; synthesis: the overlap-add sums */
    int cur;
    double *d_spectrum;                     /* size interleaved complex points */
    double *d_io;                           /* hop samples, or the two planes of bins values */
} asm1_t;

static void asm1_destroy(asm1_t *f)
{
    if (!f) return;
    llzs_free(f->d_window); llzs_free(f->d_fft_cs); llzs_free(f->d_run[0]); llzs_free(f->d_run[1]);
    llzs_free(f->d_spectrum); llzs_free(f->d_io);
    f->tag = 0;
    free(f);
}

static double *upload_f64(const double *host, size_t count)
{
    double *dev = (double *)llzs_malloc(sizeof(double) * count);
    if (dev && llzs_h2d(dev, host, sizeof(double) * count, NULL) != LLZ_OK) { llzs_free(dev); dev = NULL; }
    return dev;
}

static double *zeros_f64(size_t count)
{
    double *dev = (double *)llzs_malloc(sizeof(double) * count);
    if (dev && llzs_memset(dev, 0, sizeof(double) * count, NULL) != LLZ_OK) { llzs_free(dev); dev = NULL; }
    return dev;
}

static unsigned long asm1_init(int overlap_hint, int frame_len, win_t win_type, const char *who)
{
    int fft_len;
    double magic;
    if (!asm_shape(overlap_hint, frame_len, &fft_len, &magic) || fft_len > 4096) {
        llzs_set_error("%s: overlap_hint %d frame_len %d (fft_len must be a power of two <= 4096)", who, overlap_hint,
                       frame_len);
        return LLZ_BAD_HANDLE;
    }
    asm1_t *f = (asm1_t *)calloc(1, sizeof(*f));
    double *w = (double *)malloc(sizeof(double) * (size_t)fft_len);
    int ok = f && w;
    if (ok && !asm_window(w, fft_len, win_type)) {
        llzs_set_error("%s: unknown window %d", who, (int)win_type);
        ok = 0;
    }
    if (ok) {
        f->tag = LLZ_TAG_ASM1; f->device = llzs_device_get();
        f->hop = frame_len; f->size = fft_len; f->bins = (fft_len >> 1) + 1; f->out_scale = magic;
        f->d_window = upload_f64(w, (size_t)fft_len);
        f->d_fft_cs = llz_host_fft_table_f64(fft_len);
        f->d_run[0] = zeros_f64((size_t)fft_len);
        f->d_run[1] = zeros_f64((size_t)fft_len);
        f->d_spectrum = zeros_f64(2 * (size_t)fft_len);
        f->d_io = zeros_f64(2 * (size_t)f->bins > (size_t)frame_len ? 2 * (size_t)f->bins : (size_t)frame_len);
        ok = f->device >= 0 && f->d_window && f->d_fft_cs && f->d_run[0] && f->d_run[1] && f->d_spectrum && f->d_io &&
             llzs_sync(NULL) == LLZ_OK;
    }
    free(w);
    if (!ok) {
        if (f && f->tag) asm1_destroy(f); else free(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

static void asm1_uninit(unsigned long handle)
{
    if (!LLZ_HANDLE_OK(handle, asm1_t, LLZ_TAG_ASM1)) return;
    asm1_t *f = (asm1_t *)handle;
    const int prev = llzs_device_enter(f->device);
    llzs_sync(NULL);
    asm1_destroy(f);
    llzs_device_leave(prev);
}

unsigned long llz_analysis_fft_init(int overlap_hint, int frame_len, win_t win_type)
{
    return asm1_init(overlap_hint, frame_len, win_type, "llz_analysis_fft_init");
}

unsigned long llz_synthesis_fft_init(int overlap_hint, int frame_len, win_t win_type)
{
    return asm1_init(overlap_hint, frame_len, win_type, "llz_synthesis_fft_init");
}

void llz_analysis_fft_uninit(unsigned long handle) { asm1_uninit(handle); }
void llz_synthesis_fft_uninit(unsigned long handle) { asm1_uninit(handle); }

/* reference llz_asmodel.c:177-203: slide the input buffer by one frame, window, forward transform, bins 0..N/2 */
void llz_analysis_fft(unsigned long handle, double *x, double *re, double *im)
{
    if (!LLZ_HANDLE_OK(handle, asm1_t, LLZ_TAG_ASM1) || !x || !re || !im) {
        llzs_set_error("llz_analysis_fft: bad handle or arguments");
        return;                                                     /* void in the reference ABI */
    }
    asm1_t *f = (asm1_t *)handle;
    const size_t plane = sizeof(double) * (size_t)f->bins;
    const int prev = llzs_device_enter(f->device);
    int rc = llzs_h2d(f->d_io, x, sizeof(double) * (size_t)f->hop, NULL);
    if (rc == LLZ_OK)
        rc = llzs_frame_slide_window_f64(f->d_io, f->d_run[f->cur], f->d_run[f->cur ^ 1], f->d_window, f->d_spectrum,
                                         f->size, f->hop, 1, NULL);
    if (rc == LLZ_OK) f->cur ^= 1;
    if (rc == LLZ_OK) rc = llzs_fft_f64(f->d_spectrum, f->size, f->d_fft_cs, 0, NULL);
    if (rc == LLZ_OK) rc = llzs_spectrum_split_f64(f->d_spectrum, f->d_io, f->bins, NULL);
    if (rc == LLZ_OK) rc = llzs_d2h(re, f->d_io, plane, NULL);
    if (rc == LLZ_OK) rc = llzs_d2h(im, f->d_io + f->bins, plane, NULL);
    llzs_device_leave(prev);
}

/* reference llz_asmodel.c:274-309: conjugate-mirror the half spectrum, inverse transform, window, overlap-add, emit */
void llz_synthesis_fft(unsigned long handle, double *re, double *im, double *x)
{
    if (!LLZ_HANDLE_OK(handle, asm1_t, LLZ_TAG_ASM1) || !x || !re || !im) {
        llzs_set_error("llz_synthesis_fft: bad handle or arguments");
        return;
    }
    asm1_t *f = (asm1_t *)handle;
    const size_t plane = sizeof(double) * (size_t)f->bins;
    const int prev = llzs_device_enter(f->device);
    int rc = llzs_h2d(f->d_io, re, plane, NULL);
    if (rc == LLZ_OK) rc = llzs_h2d(f->d_io + f->bins, im, plane, NULL);
    if (rc == LLZ_OK) rc = llzs_spectrum_mirror_f64(f->d_io, f->d_spectrum, f->size, NULL);
    if (rc == LLZ_OK) rc = llzs_fft_f64(f->d_spectrum, f->size, f->d_fft_cs, 1, NULL);
    if (rc == LLZ_OK)
        rc = llzs_frame_overlap_add_f64(f->d_spectrum, 2, f->d_window, f->d_run[f->cur], f->d_run[f->cur ^ 1], f->d_io,
                                        f->size, f->hop, f->out_scale, NULL);
    if (rc == LLZ_OK) f->cur ^= 1;
    if (rc == LLZ_OK) rc = llzs_d2h(x, f->d_io, sizeof(double) * (size_t)f->hop, NULL);
    llzs_device_leave(prev);
}

/* ---- Part 1b: MDCT frames with 50 % overlap (llz_asmodel.c:313-463), same scheme around the device MDCT ---- */

#define LLZ_TAG_ASMD 0x4c5a5344

typedef struct {
    int tag, device, hop, size;             /* hop = frame_len, size = mdct_len = 2 hop */
    double *d_window;
    double *d_run[2];                       /* analysis: the last two frames; synthesis: the overlap-add sums */
    int cur;
    double *d_frame, *d_coef, *d_io;        /* size windowed samples, hop coefficients, hop samples in / out */
    unsigned long mdct;                     /* llz_mdct_init(MDCT_FFT4, size): the N/4-point form, llz_asmodel.c:323 */
} asmd_t;

static void asmd_destroy(asmd_t *f)
{
    if (!f) return;
    if (f->mdct && f->mdct != LLZ_BAD_HANDLE) llz_mdct_uninit(f->mdct);
    llzs_free(f->d_window); llzs_free(f->d_run[0]); llzs_free(f->d_run[1]);
    llzs_free(f->d_frame); llzs_free(f->d_coef); llzs_free(f->d_io);
    f->tag = 0;
    free(f);
}

static unsigned long asmd_init(int frame_len, mdct_win_t win_type, const char *who)
{
    if (frame_len < 4 || frame_len > 8192 || (frame_len & (frame_len - 1)) ||
        (win_type != MDCT_SINE && win_type != MDCT_KBD)) {
        llzs_set_error("%s: frame_len %d (a power of two in 4..8192) window %d", who, frame_len, (int)win_type);
        return LLZ_BAD_HANDLE;
    }
    const int N = frame_len << 1;
    asmd_t *f = (asmd_t *)calloc(1, sizeof(*f));
    double *w = (double *)malloc(sizeof(double) * (size_t)N);
    int ok = f && w;
    if (ok) {
        f->tag = LLZ_TAG_ASMD; f->device = llzs_device_get(); f->hop = frame_len; f->size = N;
        if (win_type == MDCT_SINE) llz_mdct_sine(w, N);
        else llz_mdct_kbd(w, N, 6);                                  /* llz_asmodel.c:330-331 */
        f->d_window = upload_f64(w, (size_t)N);
        f->d_run[0] = zeros_f64((size_t)N);
        f->d_run[1] = zeros_f64((size_t)N);
        f->d_frame = zeros_f64((size_t)N);
        f->d_coef = zeros_f64((size_t)frame_len);
        f->d_io = zeros_f64((size_t)frame_len);
        f->mdct = llz_mdct_init(MDCT_FFT4, N);
        ok = f->device >= 0 && f->d_window && f->d_run[0] && f->d_run[1] && f->d_frame && f->d_coef && f->d_io &&
             f->mdct != LLZ_BAD_HANDLE && llzs_sync(NULL) == LLZ_OK;
    }
    free(w);
    if (!ok) {
        if (f && f->tag) asmd_destroy(f); else free(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

static void asmd_uninit(unsigned long handle)
{
    if (!LLZ_HANDLE_OK(handle, asmd_t, LLZ_TAG_ASMD)) return;
    asmd_t *f = (asmd_t *)handle;
    const int prev = llzs_device_enter(f->device);
    llzs_sync(NULL);
    asmd_destroy(f);
    llzs_device_leave(prev);
}

unsigned long llz_analysis_mdct_init(int frame_len, mdct_win_t win_type)
{
    return asmd_init(frame_len, win_type, "llz_analysis_mdct_init");
}

unsigned long llz_synthesis_mdct_init(int frame_len, mdct_win_t win_type)
{
    return asmd_init(frame_len, win_type, "llz_synthesis_mdct_init");
}

void llz_analysis_mdct_uninit(unsigned long handle) { asmd_uninit(handle); }
void llz_synthesis_mdct_uninit(unsigned long handle) { asmd_uninit(handle); }

/* reference llz_asmodel.c:365-383 */
void llz_analysis_mdct(unsigned long handle, double *x, double *X)
{
    if (!LLZ_HANDLE_OK(handle, asmd_t, LLZ_TAG_ASMD) || !x || !X) {
        llzs_set_error("llz_analysis_mdct: bad handle or arguments");
        return;
    }
    asmd_t *f = (asmd_t *)handle;
    const size_t frame = sizeof(double) * (size_t)f->hop;
    const int prev = llzs_device_enter(f->device);
    int rc = llzs_h2d(f->d_io, x, frame, NULL);
    if (rc == LLZ_OK)
        rc = llzs_frame_slide_window_f64(f->d_io, f->d_run[f->cur], f->d_run[f->cur ^ 1], f->d_window, f->d_frame,
                                         f->size, f->hop, 0, NULL);
    if (rc == LLZ_OK) f->cur ^= 1;
    if (rc == LLZ_OK) rc = llz_host_mdct_on_device(f->mdct, f->d_frame, f->d_coef, 0);
    if (rc == LLZ_OK) rc = llzs_d2h(X, f->d_coef, frame, NULL);
    llzs_device_leave(prev);
}

/* reference llz_asmodel.c:440-462 */
void llz_synthesis_mdct(unsigned long handle, double *X, double *x)
{
    if (!LLZ_HANDLE_OK(handle, asmd_t, LLZ_TAG_ASMD) || !x || !X) {
        llzs_set_error("llz_synthesis_mdct: bad handle or arguments");
        return;
    }
    asmd_t *f = (asmd_t *)handle;
    const size_t frame = sizeof(double) * (size_t)f->hop;
    const int prev = llzs_device_enter(f->device);
    int rc = llzs_h2d(f->d_coef, X, frame, NULL);
    if (rc == LLZ_OK) rc = llz_host_mdct_on_device(f->mdct, f->d_coef, f->d_frame, 1);
    if (rc == LLZ_OK)
        rc = llzs_frame_overlap_add_f64(f->d_frame, 1, f->d_window, f->d_run[f->cur], f->d_run[f->cur ^ 1], f->d_io,
                                        f->size, f->hop, 1.0, NULL);
    if (rc == LLZ_OK) f->cur ^= 1;
    if (rc == LLZ_OK) rc = llzs_d2h(x, f->d_io, frame, NULL);
    llzs_device_leave(prev);
}

/* ---- Part 2: batch extension ---- */

typedef struct {
    int tag, device, channels, frame_len, fft_len, bins;
    float magic;
    float *d_w, *d_cs;
    float *d_hist[2], *d_ola[2];        /* analysis history / synthesis overlap-add tail: [channels][fft_len-frame_len] */
    int cur_hist, cur_ola;
    llz_stage_t st_x, st_re, st_im;
    void *stream;
} asmm_t;

static void asmm_destroy(asmm_t *f)
{
    if (!f) return;
    llzs_free(f->d_w); llzs_free(f->d_cs);
    llzs_free(f->d_hist[0]); llzs_free(f->d_hist[1]); llzs_free(f->d_ola[0]); llzs_free(f->d_ola[1]);
    llz_stage_release(&f->st_x); llz_stage_release(&f->st_re); llz_stage_release(&f->st_im);
    f->tag = 0;
    free(f);
}

unsigned long llz_stft_mc_init(int channels, int overlap_hint, int frame_len, win_t win_type)
{
    int N;
    double magic;
    if (channels < 1 || !asm_shape(overlap_hint, frame_len, &N, &magic) || N < 8 || N > 2048) {
        llzs_set_error("llz_stft_mc_init: channels %d overlap_hint %d frame_len %d (fft_len a power of two in 8..2048)",
                       channels, overlap_hint, frame_len);
        return LLZ_BAD_HANDLE;
    }
    asmm_t *f = (asmm_t *)calloc(1, sizeof(*f));
    double *w = (double *)malloc(sizeof(double) * (size_t)N);
    float *tab = (float *)malloc(sizeof(float) * 3 * (size_t)N);
    int rc = (f && w && tab) ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK && !asm_window(w, N, win_type)) {
        llzs_set_error("llz_stft_mc_init: unknown window %d", (int)win_type);
        rc = LLZ_ERR_ARG;
    }
    if (rc == LLZ_OK) {
        f->tag = LLZ_TAG_ASMM; f->device = llzs_device_get(); f->channels = channels; f->frame_len = frame_len; f->fft_len = N; f->bins = N / 2 + 1;
        f->magic = (float)magic;
        for (int i = 0; i < N; i++) {
            const double ang = (double)(2 * M_PI * i) / N;          /* table of llz_fft_init, llz_fft.c:222-229 */
            tab[i] = (float)w[i];
            tab[N + i] = (float)cos(ang);
            tab[2 * N + i] = (float)sin(ang);
        }
        const size_t keep = sizeof(float) * (size_t)channels * (size_t)(N - frame_len);
        f->d_w = (float *)llzs_malloc(sizeof(float) * (size_t)N);
        f->d_cs = (float *)llzs_malloc(sizeof(float) * 2 * (size_t)N);
        for (int k = 0; k < 2; k++) {
            f->d_hist[k] = (float *)llzs_malloc(keep);
            f->d_ola[k] = (float *)llzs_malloc(keep);
        }
        if (!f->d_w || !f->d_cs || !f->d_hist[0] || !f->d_hist[1] || !f->d_ola[0] || !f->d_ola[1]) rc = LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) rc = llzs_h2d(f->d_w, tab, sizeof(float) * (size_t)N, NULL);
        if (rc == LLZ_OK) rc = llzs_h2d(f->d_cs, tab + N, sizeof(float) * 2 * (size_t)N, NULL);
        for (int k = 0; k < 2 && rc == LLZ_OK; k++) {
            rc = llzs_memset(f->d_hist[k], 0, keep, NULL);
            if (rc == LLZ_OK) rc = llzs_memset(f->d_ola[k], 0, keep, NULL);
        }
        if (rc == LLZ_OK) rc = llzs_sync(NULL);
    }
    free(w); free(tab);
    if (rc != LLZ_OK) {
        if (f && f->tag) asmm_destroy(f); else free(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

void llz_stft_mc_uninit(unsigned long handle)
{
    if (LLZ_HANDLE_OK(handle, asmm_t, LLZ_TAG_ASMM)) {
        const int prev = llzs_device_enter(((asmm_t *)handle)->device);
        llzs_sync(((asmm_t *)handle)->stream);
        asmm_destroy((asmm_t *)handle);
        llzs_device_leave(prev);
    }
}

int llz_stft_mc_bins(unsigned long handle)
{
    return LLZ_HANDLE_OK(handle, asmm_t, LLZ_TAG_ASMM) ? ((asmm_t *)handle)->bins : LLZ_ERR_ARG;
}

int llz_stft_mc_set_stream(unsigned long handle, void *stream)
{
    if (!LLZ_HANDLE_OK(handle, asmm_t, LLZ_TAG_ASMM)) return LLZ_ERR_ARG;
    ((asmm_t *)handle)->stream = stream;
    return LLZ_OK;
}

/* device views of the three caller buffers: x [C][frames*F], re/im [C][frames][bins]; `in` marks which side is input */
static int asmm_stage(asmm_t *f, const float *x, const float *re, const float *im, int frames, int x_is_input,
                      float **dx, float **dre, float **dim)
{
    const size_t xb = sizeof(float) * (size_t)f->channels * frames * f->frame_len;
    const size_t sb = sizeof(float) * (size_t)f->channels * frames * f->bins;
    int rc = LLZ_OK;
    *dx = (float *)x; *dre = (float *)re; *dim = (float *)im;
    const int x_dev = llzs_is_device_ptr(x), re_dev = llzs_is_device_ptr(re), im_dev = llzs_is_device_ptr(im);
    if (x_dev < 0 || re_dev < 0 || im_dev < 0) return LLZ_ERR_ARG;
    if (!x_dev) {
        *dx = (float *)llz_stage_reserve(&f->st_x, xb);
        if (!*dx) return LLZ_ERR_NOMEM;
        if (x_is_input) rc = llzs_h2d(*dx, x, xb, f->stream);
    }
    if (rc == LLZ_OK && !re_dev) {
        *dre = (float *)llz_stage_reserve(&f->st_re, sb);
        if (!*dre) return LLZ_ERR_NOMEM;
        if (!x_is_input) rc = llzs_h2d(*dre, re, sb, f->stream);
    }
    if (rc == LLZ_OK && !im_dev) {
        *dim = (float *)llz_stage_reserve(&f->st_im, sb);
        if (!*dim) return LLZ_ERR_NOMEM;
        if (!x_is_input) rc = llzs_h2d(*dim, im, sb, f->stream);
    }
    return rc;
}

int llz_stft_mc_analysis(unsigned long handle, const float *x, float *re, float *im, int frames)
{
    if (!LLZ_HANDLE_OK(handle, asmm_t, LLZ_TAG_ASMM) || !x || !re || !im || frames < 1) {
        llzs_set_error("llz_stft_mc_analysis: bad handle or arguments");
        return LLZ_ERR_ARG;
    }
    asmm_t *f = (asmm_t *)handle;
    const long n = (long)frames * f->frame_len;
    float *dx, *dre, *dim;
    const int prev = llzs_device_enter(f->device);
    int rc = asmm_stage(f, x, re, im, frames, 1, &dx, &dre, &dim);
    if (rc == LLZ_OK)
        rc = llzs_stft_analysis_f32(dx, f->d_hist[f->cur_hist], dre, dim, f->d_w, f->d_cs, f->channels, frames,
                                    f->frame_len, f->fft_len, n, f->stream);
    /* history for the next call: the last fft_len - frame_len samples of concat(history, x) */
    if (rc == LLZ_OK)
        rc = llzs_fir_tail_f32(dx, f->d_hist[f->cur_hist], f->d_hist[f->cur_hist ^ 1], f->channels, (int)n, n,
                               f->fft_len - f->frame_len + 1, f->stream);
    if (rc == LLZ_OK) f->cur_hist ^= 1;
    const size_t sb = sizeof(float) * (size_t)f->channels * frames * f->bins;
    if (rc == LLZ_OK && dre != re) rc = llzs_d2h(re, dre, sb, f->stream);
    if (rc == LLZ_OK && dim != im) rc = llzs_d2h(im, dim, sb, f->stream);
    llzs_device_leave(prev);
    return rc == LLZ_OK ? frames : rc;
}

int llz_stft_mc_synthesis(unsigned long handle, const float *re, const float *im, float *x, int frames)
{
    if (!LLZ_HANDLE_OK(handle, asmm_t, LLZ_TAG_ASMM) || !x || !re || !im || frames < 1) {
        llzs_set_error("llz_stft_mc_synthesis: bad handle or arguments");
        return LLZ_ERR_ARG;
    }
    asmm_t *f = (asmm_t *)handle;
    const long n = (long)frames * f->frame_len;
    float *dx, *dre, *dim;
    const int prev = llzs_device_enter(f->device);
    int rc = asmm_stage(f, x, re, im, frames, 0, &dx, &dre, &dim);
    if (rc == LLZ_OK)
        rc = llzs_stft_synthesis_f32(dre, dim, dx, f->d_ola[f->cur_ola], f->d_ola[f->cur_ola ^ 1], f->d_w, f->d_cs,
                                     f->channels, frames, f->frame_len, f->fft_len, n, f->magic, f->stream);
    if (rc == LLZ_OK) f->cur_ola ^= 1;
    if (rc == LLZ_OK && dx != x) rc = llzs_d2h(x, dx, sizeof(float) * (size_t)f->channels * n, f->stream);
    llzs_device_leave(prev);
    return rc == LLZ_OK ? frames : rc;
}

/* ---- Part 2b: windowed MDCT frames (time-domain alias cancellation) in batch, float32 ---- */
/* The batch form of llz_analysis_mdct / llz_synthesis_mdct (llz_asmodel.c:313-463): per channel, analysis keeps the previous
 * frame and synthesis the overlap-add tail, so consecutive calls continue the streams.  On the register MDCT kernels
 * (k_mdct_reg_f32<..., FRAMES>): the window is applied as the transform reads / writes, frames are taken straight from the
 * planar signal at a hop of frame_len (nothing is expanded in memory). */
#define LLZ_TAG_AMDM 0x4c5a4d4d

typedef struct {
    int tag, device, channels, frame_len, mdct_len;
    float *d_w, *d_tc, *d_ts, *d_cs;
    float *d_prev[2], *d_tail[2];       /* analysis: the previous frame; synthesis: the overlap-add tail; [channels][frame_len] */
    int cur_prev, cur_tail;
    llz_stage_t st_x, st_X;
    void *stream;
} amdm_t;

static void amdm_destroy(amdm_t *f)
{
    if (!f) return;
    llzs_free(f->d_w); llzs_free(f->d_tc); llzs_free(f->d_ts); llzs_free(f->d_cs);
    llzs_free(f->d_prev[0]); llzs_free(f->d_prev[1]); llzs_free(f->d_tail[0]); llzs_free(f->d_tail[1]);
    llz_stage_release(&f->st_x); llz_stage_release(&f->st_X);
    f->tag = 0;
    free(f);
}

unsigned long llz_mdct_frames_mc_init(int channels, int frame_len, mdct_win_t win_type)
{
    if (channels < 1 || frame_len < 128 || frame_len > 4096 || (frame_len & (frame_len - 1)) ||
        (win_type != MDCT_SINE && win_type != MDCT_KBD)) {
        llzs_set_error("llz_mdct_frames_mc_init: channels %d frame_len %d (a power of two in 128..4096) window %d",
                       channels, frame_len, (int)win_type);
        return LLZ_BAD_HANDLE;
    }
    const int N = frame_len << 1, N4 = N >> 2;
    amdm_t *f = (amdm_t *)calloc(1, sizeof(*f));
    double *w = (double *)malloc(sizeof(double) * (size_t)N);
    float *tab = (float *)malloc(sizeof(float) * ((size_t)N + 4 * (size_t)N4));
    int rc = (f && w && tab) ? LLZ_OK : LLZ_ERR_NOMEM;
    if (rc == LLZ_OK) {
        f->tag = LLZ_TAG_AMDM; f->device = llzs_device_get(); f->channels = channels; f->frame_len = frame_len; f->mdct_len = N;
        if (win_type == MDCT_SINE) llz_mdct_sine(w, N); else llz_mdct_kbd(w, N, 6);     /* llz_asmodel.c:327-334 */
        float *tw = tab + N;
        for (int i = 0; i < N; i++) tab[i] = (float)w[i];
        for (int k = 0; k < N4; k++) {
            tw[k] = (float)cos(-2 * M_PI * (k + 0.125) / N);              /* llz_mdct.c:459-462 */
            tw[N4 + k] = (float)sin(-2 * M_PI * (k + 0.125) / N);
            const double ang = (double)(2 * M_PI * k) / N4;               /* llz_fft.c:222-229 for size N/4 */
            tw[2 * N4 + k] = (float)cos(ang);
            tw[3 * N4 + k] = (float)sin(ang);
        }
        const size_t keep = sizeof(float) * (size_t)channels * (size_t)frame_len;
        f->d_w = (float *)llzs_malloc(sizeof(float) * (size_t)N);
        f->d_tc = (float *)llzs_malloc(sizeof(float) * (size_t)N4);
        f->d_ts = (float *)llzs_malloc(sizeof(float) * (size_t)N4);
        f->d_cs = (float *)llzs_malloc(sizeof(float) * 2 * (size_t)N4);
        for (int k = 0; k < 2; k++) {
            f->d_prev[k] = (float *)llzs_malloc(keep);
            f->d_tail[k] = (float *)llzs_malloc(keep);
        }
        if (!f->d_w || !f->d_tc || !f->d_ts || !f->d_cs || !f->d_prev[0] || !f->d_prev[1] || !f->d_tail[0] || !f->d_tail[1])
            rc = LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) rc = llzs_h2d(f->d_w, tab, sizeof(float) * (size_t)N, NULL);
        if (rc == LLZ_OK) rc = llzs_h2d(f->d_tc, tw, sizeof(float) * (size_t)N4, NULL);
        if (rc == LLZ_OK) rc = llzs_h2d(f->d_ts, tw + N4, sizeof(float) * (size_t)N4, NULL);
        if (rc == LLZ_OK) rc = llzs_h2d(f->d_cs, tw + 2 * N4, sizeof(float) * 2 * (size_t)N4, NULL);
        for (int k = 0; k < 2 && rc == LLZ_OK; k++) {
            rc = llzs_memset(f->d_prev[k], 0, keep, NULL);
            if (rc == LLZ_OK) rc = llzs_memset(f->d_tail[k], 0, keep, NULL);
        }
        if (rc == LLZ_OK) rc = llzs_sync(NULL);
    }
    free(w); free(tab);
    if (rc != LLZ_OK) {
        if (f && f->tag) amdm_destroy(f); else free(f);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)f;
}

void llz_mdct_frames_mc_uninit(unsigned long handle)
{
    if (LLZ_HANDLE_OK(handle, amdm_t, LLZ_TAG_AMDM)) {
        const int prev = llzs_device_enter(((amdm_t *)handle)->device);
        llzs_sync(((amdm_t *)handle)->stream);
        amdm_destroy((amdm_t *)handle);
        llzs_device_leave(prev);
    }
}

int llz_mdct_frames_mc_set_stream(unsigned long handle, void *stream)
{
    if (!LLZ_HANDLE_OK(handle, amdm_t, LLZ_TAG_AMDM)) return LLZ_ERR_ARG;
    ((amdm_t *)handle)->stream = stream;
    return LLZ_OK;
}

static int amdm_run(unsigned long handle, const float *in, float *out, int frames, int inverse, const char *who)
{
    if (!LLZ_HANDLE_OK(handle, amdm_t, LLZ_TAG_AMDM) || !in || !out || frames < 1 || in == out) {
        llzs_set_error("%s: bad handle or arguments", who);
        return LLZ_ERR_ARG;
    }
    amdm_t *f = (amdm_t *)handle;
    const size_t bytes = sizeof(float) * (size_t)f->channels * (size_t)frames * (size_t)f->frame_len;   /* both sides */
    const int prev = llzs_device_enter(f->device);
    const int in_dev = llzs_is_device_ptr(in), out_dev = llzs_is_device_ptr(out);
    const float *d_in = in;
    float *d_out = out;
    int rc = (in_dev < 0 || out_dev < 0) ? LLZ_ERR_ARG : LLZ_OK;
    if (rc == LLZ_OK && !in_dev) {
        d_in = (const float *)llz_stage_reserve(inverse ? &f->st_X : &f->st_x, bytes);
        rc = d_in ? llzs_h2d((void *)d_in, in, bytes, f->stream) : LLZ_ERR_NOMEM;
    }
    if (rc == LLZ_OK && !out_dev) {
        d_out = (float *)llz_stage_reserve(inverse ? &f->st_x : &f->st_X, bytes);
        if (!d_out) rc = LLZ_ERR_NOMEM;
    }
    float **st = inverse ? f->d_tail : f->d_prev;
    int *cur = inverse ? &f->cur_tail : &f->cur_prev;
    if (rc == LLZ_OK)
        rc = llzs_mdct4_frames_f32(d_in, d_out, f->channels, frames, f->mdct_len, f->d_tc, f->d_ts, f->d_cs, f->d_w, st[*cur],
                                   st[*cur ^ 1], inverse, f->stream);
    if (rc == LLZ_OK) *cur ^= 1;
    if (rc == LLZ_OK && !out_dev) rc = llzs_d2h(out, d_out, bytes, f->stream);
    llzs_device_leave(prev);
    return rc == LLZ_OK ? frames : rc;
}

int llz_mdct_frames_mc_analysis(unsigned long handle, const float *x, float *X, int frames)
{
    return amdm_run(handle, x, X, frames, 0, "llz_mdct_frames_mc_analysis");
}

int llz_mdct_frames_mc_synthesis(unsigned long handle, const float *X, float *x, int frames)
{
    return amdm_run(handle, X, x, frames, 1, "llz_mdct_frames_mc_synthesis");
}
