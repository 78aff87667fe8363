/*
 * include/llz_pcm.h -- PCM ingest / egress on the device: the step before and after the filter
 * path. WAV data chunks are sample-interleaved int16 ([n][channels], what reference example/llz_resample/main.c:96-114
 * reads and writes for one channel); the filter kernels want planar float32 [channels][n]. Header parsing stays on
 * the host with the reference's libllzaudio/llz_wavfmt.c:82-213 -- only the bulk conversion belongs on the GPU.
 */
#ifndef LLZ_PCM_H
#define LLZ_PCM_H

#ifdef __cplusplus
extern "C" {
#endif

/* out[c][i] = (float)in[i*channels + c] * scale   (scale = 1/32768 maps int16 onto [-1, 1): exact in float32).
 * in: n*channels int16, out: planar [channels][n] float32; device or host pointers. Returns 0 or < 0. */
int llz_pcm_deinterleave_i16_f32(const short *in, float *out, int channels, long n, float scale, void *stream);
/* out[i*channels + c] = (short)clamp(in[c][i] * scale, -32768, 32767): the reference's own float->int16 rule
 * (clamp, then C truncation toward zero: llz_resample.c:596-601). */
int llz_pcm_interleave_f32_i16(const float *in, short *out, int channels, long n, float scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif
