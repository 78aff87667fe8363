/*
 * include/llz_asmodel.h -- analysis / synthesis by windowed FFT frames (overlap-add), C ABI of libllzfilter_hip.so
 * (the next caller of llz_fft and the window functions).
 * Part 1: the reference's symbols (reference libllzfilter/llz_asmodel.h:36-42, llz_asmodel.c:109-310): one frame per
 *         call on host `double` buffers, the transform on the GPU in the reference's operation order and the framing on
 *         the host in the reference's statement order: bit-identical results.
 * Part 2: many channels and frames per call, float32.
 * The MDCT half of the reference header (llz_analysis_mdct_* / llz_synthesis_mdct_*) sits on
 * llz_mdct.h the same way.
 */
#ifndef LLZ_ASMODEL_H
#define LLZ_ASMODEL_H

#include "llz_fir.h"     /* win_t */
#include "llz_mdct.h"    /* mdct_win_t */

#ifdef __cplusplus
extern "C" {
#endif

enum {
    LLZ_OVERLAP_HIGH = 0,       /* 3/4 overlap: fft_len = 4 * frame_len, output scaled by the reference's 0.812 */
    LLZ_OVERLAP_LOW             /* 1/2 overlap: fft_len = 2 * frame_len, scale 1 */
};

/* ---- Part 1: reference-identical symbols (llz_asmodel.h:36-42) ---- */
/* fft_len must be a power of two <= 4096 here (the reference's llz_fft assumes one without checking);
 * returns (unsigned long)-1 otherwise */
unsigned long llz_analysis_fft_init(int overlap_hint, int frame_len, win_t win_type);
void          llz_analysis_fft_uninit(unsigned long handle);
/* slides frame_len new samples in, windows the last fft_len samples, transforms; re/im receive bins 0..fft_len/2 */
void          llz_analysis_fft(unsigned long handle, double *x, double *re, double *im);

unsigned long llz_synthesis_fft_init(int overlap_hint, int frame_len, win_t win_type);
void          llz_synthesis_fft_uninit(unsigned long handle);
/* inverse transform of the Hermitian extension of bins 0..fft_len/2, windowed overlap-add; x receives the oldest
 * frame_len samples of the running sum times the scale (a delay of fft_len - frame_len samples against the analysis) */
void          llz_synthesis_fft(unsigned long handle, double *re, double *im, double *x);

/* windowed MDCT frames with 50 % overlap (time-domain alias cancellation), llz_asmodel.h:45-51, llz_asmodel.c:313-463:
 * the transform is MDCT_FFT4 of length 2*frame_len; analysis followed by synthesis returns the input delayed by one
 * frame.  frame_len a power of two, 4..8192. */
unsigned long llz_analysis_mdct_init(int frame_len, mdct_win_t win_type);
void          llz_analysis_mdct_uninit(unsigned long handle);
void          llz_analysis_mdct(unsigned long handle, double *x, double *X);     /* frame_len in, frame_len coefficients */
unsigned long llz_synthesis_mdct_init(int frame_len, mdct_win_t win_type);
void          llz_synthesis_mdct_uninit(unsigned long handle);
void          llz_synthesis_mdct(unsigned long handle, double *X, double *x);

/* ---- Part 2: batch extension, float32 ---- */
/* One handle serves both directions and keeps the streaming state of each per channel (analysis: the last
 * fft_len - frame_len input samples; synthesis: the overlap-add tail), so consecutive calls continue the streams.
 * fft_len a power of two in 8..2048.  Pointers may be device or host memory. */
unsigned long llz_stft_mc_init(int channels, int overlap_hint, int frame_len, win_t win_type);
void          llz_stft_mc_uninit(unsigned long handle);
int           llz_stft_mc_bins(unsigned long handle);          /* fft_len/2 + 1 */
int           llz_stft_mc_set_stream(unsigned long handle, void *stream);
/* x: planar [channels][frames*frame_len]; re, im: [channels][frames][bins].  Returns frames or < 0. */
int           llz_stft_mc_analysis(unsigned long handle, const float *x, float *re, float *im, int frames);
int           llz_stft_mc_synthesis(unsigned long handle, const float *re, const float *im, float *x, int frames);

/* Windowed MDCT frames with 50 % overlap in batch: the float32 multi-channel form of llz_analysis_mdct / llz_synthesis_mdct
 * above (reference llz_asmodel.c:313-463: MDCT_FFT4 of length 2*frame_len on w[i]*xbuf[i]; synthesis w[i]*imdct[i] overlap-
 * added).  One handle serves both directions and keeps per channel the previous input frame (analysis) and the overlap-add
 * tail (synthesis), so consecutive calls continue the streams; analysis followed by synthesis returns the input delayed by
 * one frame.  frame_len a power of two in 128..4096.  x: planar [channels][frames*frame_len]; X: [channels][frames]
 * [frame_len].  Pointers may be device or host memory (x and X distinct).  Return frames or < 0. */
unsigned long llz_mdct_frames_mc_init(int channels, int frame_len, mdct_win_t win_type);
void          llz_mdct_frames_mc_uninit(unsigned long handle);
int           llz_mdct_frames_mc_set_stream(unsigned long handle, void *stream);
int           llz_mdct_frames_mc_analysis(unsigned long handle, const float *x, float *X, int frames);
int           llz_mdct_frames_mc_synthesis(unsigned long handle, const float *X, float *x, int frames);

#ifdef __cplusplus
}
#endif
#endif
