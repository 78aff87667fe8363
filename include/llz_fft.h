/*
 * include/llz_fft.h -- radix-2 complex FFT, C ABI of libllzfilter_hip.so.
 * Part 1: reference API (reference libllzfilter/llz_fft.h:21-25, llz_fft.c:142-249): interleaved re,im doubles,
 *         in place, forward unscaled, inverse divided by N. GPU backed, same butterfly order, no FMA contraction.
 * Part 2: batched float32 transforms on device memory (the building block of the overlap-save FIR).
 */
#ifndef LLZ_FFT_H
#define LLZ_FFT_H

#ifdef __cplusplus
extern "C" {
#endif

unsigned long llz_fft_init(int size);          /* size: power of two, 2..4096 */
void          llz_fft_uninit(unsigned long handle);
void          llz_fft(unsigned long handle, double *data);    /* host pointer, 2*size doubles */
void          llz_ifft(unsigned long handle, double *data);

unsigned long llz_fft_batch_init(int size);    /* float32; size: power of two, 8..4096 */
void          llz_fft_batch_uninit(unsigned long handle);
/* data: `count` transforms back to back, each 2*size floats (re,im interleaved); device or host pointer; in place */
int           llz_fft_batch(unsigned long handle, float *data, int count);
int           llz_ifft_batch(unsigned long handle, float *data, int count);
int           llz_fft_batch_set_stream(unsigned long handle, void *stream);

#ifdef __cplusplus
}
#endif
#endif
