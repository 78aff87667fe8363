/*
 * include/llz_mdct.h -- MDCT / IMDCT, C ABI of libllzfilter_hip.so (the N/4-point-FFT caller of
 * llz_fft).
 * Part 1: the reference's symbols (reference libllzfilter/llz_mdct.h:37-44, llz_mdct.c:97-620): one frame per call on
 *         host `double` buffers; the transform (or, for MDCT_ORIGIN, the defining sums) runs on the GPU in the
 *         reference's operation order, the twiddle steps on the host in the reference's statement order:
 *         bit-identical results for all three algorithms.
 * Part 2: many frames per call, float32, the N/4-point-FFT algorithm.
 * The fixed-point variant (reference llz_mdct_fixed.c) is in llz_mdct_fixed.h.
 */
#ifndef LLZ_MDCT_H
#define LLZ_MDCT_H

#ifdef __cplusplus
extern "C" {
#endif

typedef int mdct_win_t;

enum {
    MDCT_ORIGIN = 0,    /* the defining sums, O(N^2) (llz_mdct.c:185-222); length <= 2048 here */
    MDCT_FFT,           /* one N-point FFT (llz_mdct.c:225-264); length <= 4096 */
    MDCT_FFT4           /* one N/4-point FFT (llz_mdct.c:266-353); length <= 16384 */
};

enum {
    MDCT_SINE = 0,
    MDCT_KBD
};

/* ---- Part 1: reference-identical symbols ---- */
/* len is rounded up to a power of two as in the reference (llz_mdct.c:375-379); returns (unsigned long)-1 on failure */
unsigned long llz_mdct_init(int type, int len);
void          llz_mdct_uninit(unsigned long handle);
void          llz_mdct(unsigned long handle, double *x, double *X);      /* x: len samples -> X: len/2 coefficients */
void          llz_imdct(unsigned long handle, double *X, double *x);     /* X: len/2 -> x: len (time-aliased) */
int           llz_mdct_sine(double *w, int N);                           /* sin(pi/N (n + 1/2)): llz_mdct.c:97-108 */
int           llz_mdct_kbd(double *w, int N, double alpha);              /* Kaiser-Bessel derived: llz_mdct.c:156-182 */

/* ---- Part 2: batch extension, float32, N/4-point-FFT algorithm ---- */
/* len a power of two in 32..8192.  x: [count][len], X: [count][len/2], contiguous rows, device or host pointers. */
unsigned long llz_mdct_batch_init(int len);
void          llz_mdct_batch_uninit(unsigned long handle);
int           llz_mdct_batch_set_stream(unsigned long handle, void *stream);
int           llz_mdct_batch(unsigned long handle, const float *x, float *X, int count);
int           llz_imdct_batch(unsigned long handle, const float *X, float *x, int count);

#ifdef __cplusplus
}
#endif
#endif
