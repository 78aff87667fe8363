/*
 * include/llz_mdct_fixed.h -- fixed-point MDCT / IMDCT (int32 data, Q15 tables), C ABI of libllzfilter_hip.so
 * (fixed half).  The reference's symbols (reference libllzfilter/llz_mdct_fixed.h:24-28,
 * llz_mdct_fixed.c:116-392): one frame per call on host `int` buffers.  Everything that touches the data runs on the GPU:
 * the defining sums of MDCT_FIXED_ORIGIN, the twiddle steps of the two FFT forms (four separately floored
 * (int64 * int64) >> 15 products per rotation, wrapping adds) and the bit-exact Q15 transform between them; the host
 * builds the Q15 tables and moves the frame.  Bit-identical to the reference for all three algorithms.
 * Extension of the same shape: llz_mdct_fixed_batch / llz_imdct_fixed_batch, `count` frames per call on the same handle.
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

/* ---- batch extension: `count` frames per call, rows contiguous: x [count][len], X [count][len/2].
 * Host or device pointers (device pointers are used in place, asynchronously on the handle's stream); out of place.
 * Returns count, or a negative LLZ_ERR_* code. ---- */
int           llz_mdct_fixed_batch(unsigned long handle, const int *x, int *X, int count);
int           llz_imdct_fixed_batch(unsigned long handle, const int *X, int *x, int count);
int           llz_mdct_fixed_set_stream(unsigned long handle, void *stream);
int           llz_mdct_fixed_len(unsigned long handle);                  /* the rounded-up frame length */

#ifdef __cplusplus
}
#endif
#endif
