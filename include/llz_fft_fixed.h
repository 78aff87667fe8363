/*
 * include/llz_fft_fixed.h -- fixed-point radix-2 FFT (int32 data, Q15 twiddles), C ABI of libllzfilter_hip.so.
 * Bit-exact with reference libllzfilter/llz_fft_fixed.c:61-218: four separately floored (a*b)>>15 products per
 * butterfly, forward unscaled, inverse shifted right by log2(size) once at the end. The Q15 twiddle table is
 * built on the host exactly as llz_fft_fixed.c:243-247 + llz_fft_fixed.h:42-66 and uploaded, never recomputed
 * on the device.
 */
#ifndef LLZ_FFT_FIXED_H
#define LLZ_FFT_FIXED_H

#ifdef __cplusplus
extern "C" {
#endif

/* reference API (llz_fft_fixed.h:71-75): host pointer, 2*size ints, in place */
unsigned long llz_fft_fixed_init(int size);    /* power of two, 2..4096 */
void          llz_fft_fixed_uninit(unsigned long handle);
void          llz_fft_fixed(unsigned long handle, int *data);
void          llz_ifft_fixed(unsigned long handle, int *data);

/* batched: `count` transforms back to back in device (or host) memory */
int           llz_fft_fixed_batch(unsigned long handle, int *data, int count);
int           llz_ifft_fixed_batch(unsigned long handle, int *data, int count);
int           llz_fft_fixed_set_stream(unsigned long handle, void *stream);

#ifdef __cplusplus
}
#endif
#endif
