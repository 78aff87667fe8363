/*
 * include/llz_mdct_fixed.h -- fixed-point MDCT / IMDCT (int32 data, Q15 tables), C ABI of libllzfilter_hip.so
 * (fixed half).  The reference's symbols (reference libllzfilter/llz_mdct_fixed.h:24-28,
 * llz_mdct_fixed.c:116-392): one frame per call on host `int` buffers; the transforms run on the GPU through
 * llz_fft_fixed / llz_ifft_fixed (bit-exact Q15 butterflies), the defining sums of MDCT_FIXED_ORIGIN in a device kernel,
 * the twiddle steps on the host with the reference's macros: bit-identical results for all three algorithms.
 */
#ifndef LLZ_MDCT_FIXED_H
#define LLZ_MDCT_FIXED_H

#ifdef __cplusplus
extern "C" {
#endif

enum {
    MDCT_FIXED_ORIGIN = 0,  /* length <= 2048 here */
    MDCT_FIXED_FFT,         /* length <= 4096 */
    MDCT_FIXED_FFT4         /* length <= 16384 */
};

/* len is rounded up to a power of two as in the reference (llz_mdct_fixed.c:296-300); (unsigned long)-1 on failure */
unsigned long llz_mdct_fixed_init(int type, int len);
void          llz_mdct_fixed_uninit(unsigned long handle);
void          llz_mdct_fixed(unsigned long handle, int *x, int *X);      /* x: len -> X: len/2 */
void          llz_imdct_fixed(unsigned long handle, int *X, int *x);     /* X: len/2 -> x: len */

#ifdef __cplusplus
}
#endif
#endif
