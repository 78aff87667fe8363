/*
 * include/llz_resample.h -- decimate / interpolate / rational L/M resample, C ABI of libllzfilter_hip.so.
 * Part 1: the reference's int16 single-channel API (reference libllzfilter/llz_resample.h:32-52), GPU backed,
 *         bit-exact (double accumulate in the reference's order, clamp, truncate toward zero).
 * Part 2: multi-channel batch extension: float32 and int16 sample formats, planar [channels][n].
 */
#ifndef LLZ_RESAMPLE_H
#define LLZ_RESAMPLE_H

#include "llz_fir.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LLZ_DEFAULT_FRAMELEN 1024              /* reference llz_resample.h:32 */
#define LLZ_FRAMELEN_MAX     (160*147+8192)    /* :33 */
#define LLZ_RATIO_MAX        16                /* :35 */

/* ---- Part 1: reference-identical symbols (llz_resample.c:271-617) ---- */
unsigned long llz_decimate_init(int M, double gain, win_t win_type);
void          llz_decimate_uninit(unsigned long handle);
unsigned long llz_interp_init(int L, double gain, win_t win_type);
void          llz_interp_uninit(unsigned long handle);
unsigned long llz_resample_filter_init(int L, int M, double gain, win_t win_type);   /* (unsigned long)-1 if ratio > 16 */
void          llz_resample_filter_uninit(unsigned long handle);
int llz_get_resample_framelen_bytes(unsigned long handle);
/* sample_in: host int16 PCM, exactly llz_get_resample_framelen_bytes() bytes (else -1 where the reference
 * asserts); returns 0 and sets *sample_out_size */
int llz_decimate(unsigned long handle, unsigned char *sample_in, int sample_in_size,
                 unsigned char *sample_out, int *sample_out_size);
int llz_interp(unsigned long handle, unsigned char *sample_in, int sample_in_size,
               unsigned char *sample_out, int *sample_out_size);
int llz_resample(unsigned long handle, unsigned char *sample_in, int sample_in_size,
                 unsigned char *sample_out, int *sample_out_size);

/* ---- Part 2: multi-channel rational resampler ---- */
enum { LLZ_PCM_F32 = 0, LLZ_PCM_I16 = 1, LLZ_PCM_I16_FAST = 2 };

/* Same prototype design, tap matrix g[l][k] = L*h[kL + (lM mod L)] and indexing as llz_resample
 * (llz_resample.c:193-255, 583-603):  y[i] = gain * sum_{k<Q} x[(i*M)/L - k] * g[i mod L][k].
 * LLZ_PCM_I16: the reference's own format and result, bit for bit per channel (double accumulate in its order, clamp,
 *   truncate).  Computed by an exact integer screen on the int8 matrix cores that decides every output whose value does not
 *   lie within ~1e-5 of an integer, and by the reference's double loop for the rest (any L/M; the all-double kernel where the
 *   taps or the frame do not suit the screen).
 * LLZ_PCM_F32: float32 in/out, float accumulate, no clamp.
 * LLZ_PCM_I16_FAST: int16 in/out like LLZ_PCM_I16 (same clamp and truncation) but the sum runs in float32 on the matrix
 *   cores: a sample differs from the reference by at most one LSB (when its exact value lies within ~0.01 of an integer),
 *   RMS deviation <= 1e-5 of full scale.  Decimators only (L == 1); init fails otherwise.  (Since round 3 the bit-exact
 *   LLZ_PCM_I16 form is the faster of the two on BASELINE config 5.) */
unsigned long llz_resample_mc_init(int channels, int L, int M, double gain, win_t win_type, int pcm_format);
void          llz_resample_mc_uninit(unsigned long handle);
int  llz_resample_mc_sub_len(unsigned long handle);                 /* Q */
long llz_resample_mc_out_len(unsigned long handle, long n_in);      /* n_in*L/M; n_in*L must divide by M, else -1 */
/* in: planar [channels][n_in]; out: planar [channels][n_in*L/M]; device or host pointers of the handle's
 * sample format. History (Q-1 samples per channel) carries across calls. Returns outputs per channel or <0. */
long llz_resample_mc(unsigned long handle, const void *in, long n_in, void *out);
int  llz_resample_mc_set_stream(unsigned long handle, void *stream);
/* the L x Q tap matrix in float32 (what LLZ_PCM_F32 uses), for broadcast to other ranks: copies to host dst */
int  llz_resample_mc_get_matrix(unsigned long handle, double *dst, int capacity);
/* adopt a matrix received from another rank instead of the locally designed one (RCCL tap broadcast) */
int  llz_resample_mc_set_matrix(unsigned long handle, const double *src, int count);

#ifdef __cplusplus
}
#endif
#endif
