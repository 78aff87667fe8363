/*
 * oracle/llz_oracle.h -- CPU restatement of the libllzfilter FIR / IIR / resample / FFT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under llzlab_amd/ (the product) may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit (double / int16 / int32) against
 * the reference's own C sources compiled in this container (oracle/_ref/libllzref.so, recipe in
 * oracle/Makefile) and against the committed fixtures in tests/golden/ generated from that build
 * (oracle/gen_golden.py).  The reference has no tests or golden vectors of its own.
 *
 * All symbols carry the orc_ prefix so the oracle can be loaded next to the product library, which
 * exports the reference's llz_* names.
 */
#ifndef LLZ_ORACLE_H
#define LLZ_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_HAMMING = 0, ORC_BLACKMAN = 1, ORC_KAISER = 2 };
enum { ORC_LPF = 0, ORC_HPF = 1, ORC_BPF = 2, ORC_BSF = 3 };

/* windows and tap-count estimators: reference libllzfilter/llz_fir.c:61-193 */
int    orc_hamming(double *w, int n);
int    orc_blackman(double *w, int n);
int    orc_kaiser(double *w, int n);
int    orc_kaiser_beta(double *w, int n, double beta);
double orc_kaiser_atten2beta(double atten);
int    orc_hamming_cof_num(double ftrans);
int    orc_blackman_cof_num(double ftrans);
int    orc_kaiser_cof_num(double ftrans, double atten);

/* windowed-sinc design, reference llz_fir.c:201-393.  Writes at most n+1 taps into h (caller buffer),
 * returns the tap count actually produced (hpf/bpf/bsf force it odd). */
int    orc_fir_design(int kind, double *h, int n, double fc1, double fc2, int win);

/* dot product of llz_fir.c:411-426: x points at the newest sample */
double orc_conv(const double *x, const double *h, int h_len);

/* streaming FIR, reference llz_fir.c:442-625.  kind selects the *_init flavour. */
void  *orc_fir_new(int kind, int frame_len, int flt_len, double fc1, double fc2, int win);
void  *orc_fir_new_taps(int frame_len, const double *h, int flt_len);   /* same state machine, given taps */
int    orc_fir_flt_len(void *f);
const double *orc_fir_taps(void *f);
int    orc_fir_run(void *f, const double *in, double *out, int frame_len);
int    orc_fir_flush(void *f, double *out);
void   orc_fir_free(void *f);

/* direct-form-I IIR, reference llz_iir.c:37-156 */
void  *orc_iir_new(int M, const double *a, int N, const double *b);
int    orc_iir_run(void *f, const double *x, double *y, int frame_len);
int    orc_iir_flush(void *f, double *y);
void   orc_iir_free(void *f);

/* decimate / interp / rational resample, reference llz_resample.c:124-617.  mode 0=decimate 1=interp 2=resample */
void  *orc_rs_new(int mode, int L, int M, double gain, int win);       /* NULL when the ratio is refused */
int    orc_rs_bytes_in(void *r);
int    orc_rs_bytes_out(void *r);
int    orc_rs_num_taps(void *r);      /* prototype length n */
int    orc_rs_sub_len(void *r);       /* Q (resample) or k (polyphase) */
const double *orc_rs_matrix(void *r); /* L x Q (resample) or m x k (polyphase), row major */
const double *orc_rs_proto(void *r);  /* prototype low-pass h[n] */
int    orc_rs_run(void *r, const unsigned char *in, int in_bytes, unsigned char *out, int *out_bytes);
void   orc_rs_free(void *r);

/* radix-2 complex FFT on interleaved doubles, reference llz_fft.c:33-249 */
void  *orc_fft_new(int size);
void   orc_fft_fwd(void *f, double *data);
void   orc_fft_inv(void *f, double *data);
void   orc_fft_free(void *f);

/* fixed-point FFT (int32 data, Q15 twiddles), reference llz_fft_fixed.c:33-268 */
void  *orc_fftx_new(int size);
const short *orc_fftx_cos(void *f);
const short *orc_fftx_sin(void *f);
void   orc_fftx_fwd(void *f, int *data);
void   orc_fftx_inv(void *f, int *data);
void   orc_fftx_free(void *f);

/* correlation, reference libllzfilter/llz_corr.c:38-177 */
void   orc_autocorr(const double *x, int n, int p, double *r);
void   orc_crosscorr(const double *x, const double *y, int n, int p, double *r);
double orc_corr_cof(const double *a, const double *b, int len);
void  *orc_acf_new(int n);                       /* FFT autocorrelation handle: fft_len = 2^ceil(log2(2n)) */
int    orc_acf_fft_len(void *h);
void   orc_acf_run(void *h, const double *x, int n, int p, double *r);
void   orc_acf_free(void *h);

/* windowed-FFT analysis / synthesis frames, reference libllzfilter/llz_asmodel.c:109-310.
 * overlap_hint 0 = 3/4 overlap (fft_len = 4 frame_len), 1 = 1/2 overlap (fft_len = 2 frame_len) */
void  *orc_stft_new(int overlap_hint, int frame_len, int win);
int    orc_stft_fft_len(void *h);
void   orc_stft_analysis(void *h, const double *x, double *re, double *im);      /* re, im: fft_len/2 + 1 */
void   orc_stft_synthesis(void *h, const double *re, const double *im, double *x);
void   orc_stft_free(void *h);

/* MDCT, reference libllzfilter/llz_mdct.c:97-620. type 0 = defining sums, 1 = N-point FFT,
 * 2 = N/4-point FFT; size is rounded up to a power of two as in llz_mdct_init */
int    orc_mdct_sine(double *w, int N);
int    orc_mdct_kbd(double *w, int N, double alpha);
void  *orc_mdct_new(int type, int size);
int    orc_mdct_length(void *h);
void   orc_mdct_fwd(void *h, const double *x, double *X);          /* x: length, X: length/2 */
void   orc_mdct_inv(void *h, const double *X, double *x);
void   orc_mdct_free(void *h);

/* fixed-point MDCT (int32 data, Q15 tables), reference libllzfilter/llz_mdct_fixed.c:116-392; types as above */
void  *orc_mdctx_new(int type, int size);
int    orc_mdctx_length(void *h);
void   orc_mdctx_fwd(void *h, const int *x, int *X);
void   orc_mdctx_inv(void *h, const int *X, int *x);
void   orc_mdctx_free(void *h);

/* windowed MDCT frames with 50 % overlap (TDAC), reference llz_asmodel.c:313-463; win 0 = sine, 1 = KBD(alpha 6) */
void  *orc_amdct_new(int frame_len, int win);
void   orc_amdct_analysis(void *h, const double *x, double *X);    /* x: frame_len in, X: frame_len coefficients */
void   orc_amdct_synthesis(void *h, const double *X, double *x);
void   orc_amdct_free(void *h);

/* ---- batch drivers over the restatement (what the multi-channel GPU path is compared with) ---- */

/* planar [C][n] float input -> planar double output; every channel runs its own orc_fir state machine
 * with frame_len = n (one call), taps shared.  history: optional [C][flt_len-1] doubles carried in/out
 * (NULL = zero start, not returned). */
void   orc_fir_batch_f32(const float *in, double *out, int channels, long n, const double *h, int flt_len);

/* cascade of `stages` second-order sections run as chained orc_iir handles (M=N=2 each);
 * coef = stages x 6 doubles {b0,b1,b2,a0,a1,a2}. */
void   orc_iir_cascade_batch_f32(const float *in, double *out, int channels, long n,
                                 const double *coef, int stages);

/* rational resampler on float samples: same indexing and tap matrix as llz_resample (llz_resample.c:583-603),
 * double accumulate, result = gain * sum with NO clamp/truncate (float PCM has no int16 range).
 * n_in must be a multiple of M/gcd... caller guarantees n_in*L % M == 0. Returns outputs per channel. */
long   orc_rs_batch_f32(const float *in, double *out, int channels, long n_in, int L, int M,
                        double gain, int win);

/* int16 resampler over many channels and frames: runs orc_rs_run frame by frame per channel */
long   orc_rs_batch_i16(const short *in, short *out, int channels, long n_in, int L, int M,
                        double gain, int win);

/* synthetic PCM generator shared with the device generator:
 * u = fmix32(seed ^ c*0x9E3779B9 ^ n*0x85EBCA6B) */
void   orc_synth_f32(float *dst, int channels, long n, unsigned seed, int chan0);
void   orc_synth_i16(short *dst, int channels, long n, unsigned seed, int chan0);

/* PCM ingest / egress (sample-interleaved int16 <-> planar float32) and the WAV header walk of
 * libllzaudio/llz_wavfmt.c:82-213 on a memory image.  orc_wav_parse out[7]: format, channels, samplerate, bytes per
 * sample, block_align, frames, data offset */
void   orc_pcm_deinterleave_i16_f32(const short *in, float *out, int channels, long n, float scale);
void   orc_pcm_interleave_f32_i16(const float *in, short *out, int channels, long n, float scale);
int    orc_wav_parse(const unsigned char *bytes, long len, long *out);
void   orc_wav_header(unsigned char *h44, int channels, long samplerate, int bytes_per_sample, long frames);

#ifdef __cplusplus
}
#endif
#endif
