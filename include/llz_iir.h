/*
 * include/llz_iir.h -- IIR filters, C ABI of libllzfilter_hip.so.
 * Part 1: the reference's direct-form-I single-channel `double` API (reference libllzfilter/llz_iir.h:24-27).
 * Part 2: multi-channel float32 cascade of second-order sections, fused in one kernel (SURVEY.md M4:
 * an "8-biquad cascade" is 8 chained reference handles with M=N=2).
 * Part 3: multi-channel float32 form of the general direct-form-I filter itself (any orders up to 8).
 */
#ifndef LLZ_IIR_H
#define LLZ_IIR_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Part 1: reference-identical symbols (llz_iir.c:37-156) ---- */
/* a[0..M] poles (a[0] ignored, taken as 1), b[0..N] zeros (NULL = all zero). Host pointers. */
unsigned long llz_iir_filter_init(int M, double *a, int N, double *b);
void          llz_iir_filter_uninit(unsigned long handle);
int           llz_iir_filter(unsigned long handle, double *x, double *y, int frame_len);   /* returns frame_len */
int           llz_iir_filter_flush(unsigned long handle, double *y);                       /* N more samples of x=0; returns N */

/* ---- Part 2: multi-channel float32 biquad cascade ---- */
/* coef: HOST pointer, stages x 6 doubles {b0,b1,b2,a0,a1,a2} (a0 ignored), shared by all channels.
 * State (2 x + 2 y values per stage and channel) is kept in double on the device between calls. */
unsigned long llz_iir_cascade_mc_init(int channels, int stages, const double *coef);
void          llz_iir_cascade_mc_uninit(unsigned long handle);
/* planar [channels][frame_len] float32, device or host pointers; any frame_len >= 1. Returns frame_len. */
int           llz_iir_cascade_mc(unsigned long handle, const float *x, float *y, int frame_len);
int           llz_iir_cascade_mc_set_stream(unsigned long handle, void *stream);
/* working precision of the pipelined kernel for this coefficient set: 64, or 32 when every section's rounding-noise gain
 * (sum of squares of the impulse response of 1/A(z), measured at init) is at most 16 -- poles of radius up to about 0.8 */
int           llz_iir_cascade_mc_precision(unsigned long handle);

/* ---- Part 3: multi-channel GENERAL direct form I -- llz_iir_filter itself for many channels (llz_iir.c:103-156): any pole
 * order M and zero order N up to 8 (the reference's only in-tree caller uses order 3: libllzaudio/llz_musicpitch.c:1277-1285),
 * a[0..M] (a[0] ignored, taken as 1) and b[0..N] (NULL = zeros) HOST pointers shared by all channels.  float32 in and out,
 * double arithmetic in the reference's operation order; delay lines stay on the device between calls. ---- */
unsigned long llz_iir_mc_init(int channels, int M, const double *a, int N, const double *b);
void          llz_iir_mc_uninit(unsigned long handle);
/* planar [channels][frame_len] float32, device or host pointers, out of place; any frame_len >= 1. Returns frame_len. */
int           llz_iir_mc(unsigned long handle, const float *x, float *y, int frame_len);
int           llz_iir_mc_flush(unsigned long handle, float *y);         /* N more samples of x = 0 per channel: [channels][N]; returns N */
int           llz_iir_mc_set_stream(unsigned long handle, void *stream);

#ifdef __cplusplus
}
#endif
#endif
