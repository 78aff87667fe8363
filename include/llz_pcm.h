/*
 * include/llz_pcm.h -- PCM ingest / egress on the device: the step before and after the filter
 * path. WAV data chunks are sample-interleaved int16 ([n][channels], what reference example/llz_resample/main.c:96-114
 * reads and writes for one channel); the filter kernels want planar float32 [channels][n].  The WAV header (any number of
 * channels) is parsed on the host by llz_wav_parse -- the chunk walk of the reference's libllzaudio/llz_wavfmt.c:82-159 on
 * a memory image, with error returns where the reference prints and exits -- and llz_wav_ingest_f32 hands the data chunk to
 * the device kernel.
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

/* ---- WAV container (libllzaudio/llz_wavfmt.h: llz_wavfmt_t, llz_wavfmt_readheader, llz_wavfmt_writeheader) ---- */
typedef struct {
    int  format;             /* 1 = PCM, the only one accepted (llz_wavfmt.c:118-121) */
    int  channels;
    long samplerate;
    int  bytes_per_sample;   /* (bits + 7) / 8 */
    int  block_align;        /* bytes_per_sample * channels, recomputed as the reference does (:136) */
    long frames;             /* data bytes / block_align: the reference's data_size field (:155) */
    long data_offset;        /* offset of the first sample in the image */
} llz_wav_info;

/* parse a RIFF/WAVE image held in memory; LLZ_OK, or LLZ_ERR_ARG (llz_hip_last_error() says why) */
int llz_wav_parse(const unsigned char *image, long len, llz_wav_info *info);
/* the 44-byte header llz_wavfmt_writeheader emits (:185-213) */
int llz_wav_write_header(unsigned char header[44], const llz_wav_info *info);
/* whole file in memory -> planar float32 [channels][frames] (device or host pointer), samples scaled by 1/32768: 16-bit
 * PCM with any channel count; the frames that actually lie inside the image (a truncated file yields fewer).  Returns the
 * frame count or a negative code. */
long llz_wav_ingest_f32(const unsigned char *image, long len, float *planar_out, long out_capacity_frames,
                        llz_wav_info *info, void *stream);

#ifdef __cplusplus
}
#endif
#endif
