/*
 * include/llz_corr.h -- auto / cross correlation, C ABI of libllzfilter_hip.so (the first caller
 * of llz_fft + llz_ifft back to back with a pointwise step in between, the same shape as the overlap-save FIR).
 * Part 1: the reference's symbols (reference libllzfilter/llz_corr.h, llz_corr.c:38-177), host `double` buffers, computed
 *         on the GPU in the reference's operation order: bit-identical results.
 * Part 2: many frames at once, float32, planar [frames][n] -> [frames][p+1].
 */
#ifndef LLZ_CORR_H
#define LLZ_CORR_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Part 1: reference-identical symbols ---- */
/* r[k] = sum_{i} x[i]*x[i+k], k = 0..p (r has p+1 entries): llz_corr.c:38-47 */
void   llz_autocorr(double *x, int n, int p, double *r);
/* r[k] = sum_{i} x[i]*y[i+k]: llz_corr.c:49-58 */
void   llz_crosscorr(double *x, double *y, int n, int p, double *r);
/* <a,b> / sqrt(<a,a><b,b>): llz_corr.c:61-78 */
double llz_corr_cof(double *a, double *b, int len);
/* FFT autocorrelation, llz_corr.c:99-177: fft_len = 2^ceil(log2(2n)); NOTE the reference squares only the first n
 * spectrum bins and doubles the result -- kept, it is the reference's definition of this function. n <= 2048. */
unsigned long llz_autocorr_fast_init(int n);
void          llz_autocorr_fast_uninit(unsigned long handle);
void          llz_autocorr_fast(unsigned long handle, double *x, int n, int p, double *r);

/* ---- Part 2: batch extension, float32 ---- */
/* direct form for `frames` independent frames of n samples: r[f][k] = sum_i x[f][i]*x[f][i+k], k = 0..p.
 * x: planar [frames][n], r: planar [frames][p+1]; device or host pointers; p < n, p <= 255. Returns 0 or < 0. */
int llz_autocorr_mc(const float *x, float *r, int frames, int n, int p, void *stream);
/* FFT form with the reference's definition (first n bins, doubled), n <= 2048 */
unsigned long llz_autocorr_fast_mc_init(int frames, int n);
void          llz_autocorr_fast_mc_uninit(unsigned long handle);
int           llz_autocorr_fast_mc(unsigned long handle, const float *x, float *r, int p);
int           llz_autocorr_fast_mc_set_stream(unsigned long handle, void *stream);

#ifdef __cplusplus
}
#endif
#endif
