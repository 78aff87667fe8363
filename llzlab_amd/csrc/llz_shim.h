/*
 * llz_shim.h -- INTERNAL boundary between the host C layer (csrc/host, .c files, gcc) and the HIP translation
 * units (csrc/kernels, .hip files, hipcc).  "Host code stays C and calls HIP through a thin C-ABI shim": the host
 * layer never includes a HIP header; everything it needs from the device is one of these functions.
 * All pointers named dev_* / in / out / hist are DEVICE pointers unless stated; stream is a hipStream_t as void*.
 * Every function returns 0 on success or a negative LLZ_ERR_* code and records text for llz_hip_last_error().
 */
#ifndef LLZ_SHIM_H
#define LLZ_SHIM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime ---- */
void  llzs_set_error(const char *fmt, ...);
void *llzs_malloc(size_t bytes);
void  llzs_free(void *p);
int   llzs_h2d(void *dev_dst, const void *host_src, size_t bytes, void *stream);   /* synchronous for the caller */
int   llzs_d2h(void *host_dst, const void *dev_src, size_t bytes, void *stream);   /* synchronous for the caller */
int   llzs_d2d(void *dev_dst, const void *dev_src, size_t bytes, void *stream);    /* asynchronous on stream */
int   llzs_memset(void *dev_dst, int value, size_t bytes, void *stream);
int   llzs_sync(void *stream);
int   llzs_is_device_ptr(const void *p);

/* ---- tuning overrides (measurement and tests only) ----
 * The library never reads the environment.  Every algorithm / launch-shape choice is made from the problem's shape; a
 * choice can be overridden for A/B measurements and for tests that force a rarely taken form through the public
 * llz_hip_tune(name, value) (include/llz_hip.h).  value < 0 = unset (the library's own choice). */
enum {
    LLZS_TUNE_OLS_WG_PER_CU = 0,        /* resident workgroups per CU the overlap-save grid is sized for */
    LLZS_TUNE_RS_GENERIC,           /* 1: general L/M resampler without the register-window kernel; 2: first LDS kernel */
    LLZS_TUNE_RS_TILES,             /* tiles per workgroup of the general L/M resampler */
    LLZS_TUNE_RS_DEC_VALU,          /* 1: L = 1 float32 decimator on the vector pipe (LDS polyphase kernel) */
    LLZS_TUNE_RS_I16_PATH,          /* 1: no screened pass: LLZ_PCM_I16 on the all-double kernel, LLZ_PCM_I16_FAST on the float32-sum one */
    LLZS_TUNE_MFMA_NACC,            /* accumulator tiles per wave of the matrix-core FIR (1 / 2) */
    LLZS_TUNE_MFMA_WG_PER_CU,
    LLZS_TUNE_FFT_GENERIC,          /* 1: staged LDS passes instead of the register transforms */
    LLZS_TUNE_IIR_SEGS,             /* time segments per channel */
    LLZS_TUNE_IIR_UNPACKED,         /* 2: the fetch-ahead kernels with 16 samples per lane only (no 32-sample forms); (1 selected
                                     * the first wave-autonomous kernels, retired in round 3: profiles/r02/time_iir.txt keeps
                                     * their numbers) */
    LLZS_TUNE_IIR_F64,              /* 1: double arithmetic whatever the noise-gain check says */
    LLZS_TUNE_IIR_PIPE,             /* 1: stage pipeline even where the wave form would be taken */
    LLZS_TUNE_IIR_WAVE_MIN_ITEMS,   /* crossover (channel, segment) item count of the wave form */
    LLZS_TUNE_SHARD_RCCL,           /* 1: sharded handles broadcast their tables through RCCL even on a single device */
    LLZS_TUNE_RS_MFMA_FORM,         /* 1: matrix-core L/M resampler with a wave per PERIOD tile (first form) */
    LLZS_TUNE_OLS_SEG_LEN,          /* jobs per segment of the 1024-point overlap-save walk (1..16) */
    LLZS_TUNE_MDCT_RUN,             /* MDCT-frames synthesis: output segments per group (0: a launch for the even and one for
                                     * the odd frames) */
    LLZS_TUNE_MDCTQ_STEPS,          /* 1: fixed-point MDCT (both FFT forms) as three launches (step, transform, step) */
    LLZS_TUNE_RS_I16_TILES,         /* bit-exact int16 L/M resampler: period tiles per span (1..4) */
    LLZS_TUNE_RS_I16_WALK,          /* ... consecutive spans per workgroup */
    LLZS_TUNE_ACF_LDS,              /* 1: direct autocorrelation always on the LDS-window kernel (no register form for p <= 32) */
    LLZS_TUNE_STFT_FULL,            /* 1: STFT synthesis frames of 512 / 2048 points on the full-size complex inverse transform */
    LLZS_TUNE_COUNT
};
int llzs_tune(int id);                                   /* current override or -1 */

/* ---- devices, streams, events (sharded handles: llz_shard_host.c) ---- */
int   llzs_device_get(void);                       /* current device, negative on error */
int   llzs_device_set(int device);
int   llzs_device_enter(int device);               /* make `device` current; returns the previous one (or -1) */
void  llzs_device_leave(int previous);
void *llzs_stream_create(void);                    /* non-blocking stream on the current device, NULL on failure */
void  llzs_stream_destroy(void *stream);
void *llzs_event_create(void);
void  llzs_event_destroy(void *event);
int   llzs_event_record(void *event, void *stream);
double llzs_event_elapsed_ms(void *start, void *stop);   /* synchronises on stop; negative on error */

/* Coefficient-table uploads go through llzs_h2d_table so that a sharded init can see them: mode 1 uploads and records
 * (destination, size) of every table of the handle being built on this thread, mode 2 records WITHOUT uploading (the
 * tables of the other shards are filled by llzs_tables_broadcast from shard 0's), mode 0 = plain upload. */
typedef struct {
    void *dev;
    size_t bytes;
} llzs_table_ref;
#define LLZS_MAX_TABLES 16
void llzs_table_capture(int mode);
int  llzs_table_captured(llzs_table_ref *dst, int capacity);     /* number recorded since the last llzs_table_capture */
int  llzs_h2d_table(void *dev_dst, const void *host_src, size_t bytes);
/* tables[s][t]: table t of shard s (same count and sizes in every shard), shard 0 holds the data.  Shards on shard 0's
 * device get device-to-device copies; every other device receives ONE ncclBroadcast per table over a communicator of
 * the distinct devices (ncclCommInitAll, librccl loaded on first use) and hands copies to its further shards. */
int  llzs_tables_broadcast(llzs_table_ref *const *tables, int ntables, int nshards, const int *device,
                           void *const *stream);
int  llzs_tables_broadcast_ranks(void);     /* ranks of the RCCL communicator the last broadcast of this thread used (0: none) */

/* ---- FIR ---- */
#define LLZS_FIR_TAP_PAD 8      /* time-domain kernels read taps in groups of 8: pad the table with zeros */
#define LLZS_OLS_NFFT   1024
#define LLZS_OLS_MAX_TAPS 257   /* overlap = 256, 768 new samples per 1024-point block */

/* y[c][i] = sum_k taps[k] * x[c][i-k]; x[c][i<0] = hist[c][flt_len-1+i] (hist NULL = zeros).
 * in/out planar with row pitch in_pitch/out_pitch elements. taps: flt_len floats padded to a multiple of 8. */
int llzs_fir_td_f32(const float *in, float *out, const float *hist, const float *taps_padded,
                    int channels, int n, long in_pitch, long out_pitch, int flt_len, void *stream);
int llzs_fir_td_f32_fits(int flt_len);      /* 1 when the taps fit the time-domain kernel's LDS tile */
/* overlap-save: hfreq = 1024 complex floats, FFT(taps)/1024 in natural bin order; twid = 32x32 complex
 * W_1024^(a*b).  Requires flt_len <= 257. */
int llzs_fir_ols_f32(const float *in, float *out, const float *hist, const float *hfreq, const float *twid,
                     int channels, int n, long in_pitch, long out_pitch, int flt_len, void *stream);
/* overlap-save with 2048-point transforms split over the two half-waves of a wave (fir_ols.hip), 2..1025 taps: hfreq2 [2][1024] complex =
 * even then odd bins of DFT_2048(taps) / 2048, twid [32][32] W_1024^(ab), tw2k [1024] W_2048^n */
int llzs_fir_ols2k_f32(const float *in, float *out, const float *hist, const float *hfreq2, const float *twid,
                       const float *tw2k, int channels, int n, long in_pitch, long out_pitch, int flt_len, void *stream);
#define LLZS_OLS2K_MAX_TAPS 1025
/* 4096-point transforms on a whole wave (fir_ols.hip: radix-2 step + two 2048-point problems), 2..3073 taps (overlap 512 / 1024 / 2048 / 3072 by tap count): hfreq4 [4][1024]
 * complex = bins 4m + j of DFT_4096(taps) / 4096; twid [32][32] W_1024^(ab); tw2k [1024] W_2048^n; tw4k [2048] W_4096^n */
int llzs_fir_ols4k_f32(const float *in, float *out, const float *hist, const float *hfreq4, const float *twid,
                       const float *tw2k, const float *tw4k, int channels, int n, long in_pitch, long out_pitch, int flt_len,
                       void *stream);
#define LLZS_OLS4K_MAX_TAPS 3073
/* 8192-point transforms on a pair of waves (fir_ols.hip: radix-2 step across the pair + the 4096-point problem per wave), 2..6145
 * taps (overlap 1536 ... 6144 by tap count): hfreq8 [8][1024] complex = bins 8m + j of DFT_8192(taps) / 8192; twid, tw4k as above */
int llzs_fir_ols8k_f32(const float *in, float *out, const float *hist, const float *hfreq8, const float *twid,
                       const float *tw4k, int channels, int n, long in_pitch, long out_pitch, int flt_len, void *stream);
#define LLZS_OLS8K_MAX_TAPS 6145
/* time domain on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), with optional decimation:
 * y[c][i] = gain * sum_{k<T} taps[k] * x[c][i*M - k], x[c][<0] = hist[c][T-1+idx] (hist NULL = zeros); n_out outputs
 * per channel from n_in inputs, (n_out-1)*M < n_in.  taps: T floats (no padding needed). */
int llzs_fir_mfma_f32(const float *in, float *out, const float *hist, const float *taps, int channels,
                      long n_in, long n_out, long in_pitch, long out_pitch, int T, int M, float gain, void *stream);
int llzs_fir_mfma_f32_fits(int T, int M);           /* 1 when the LDS image of one tile fits */
/* the same with int16 samples in and out (fp32 accumulate, clamp, truncate toward zero): within 1 LSB of the reference's
 * double accumulation, not bit-exact; taps as floats, gain folded into them */
int llzs_fir_mfma_i16(const short *in, short *out, const short *hist, const float *taps, int channels,
                      long n_in, long n_out, long in_pitch, long out_pitch, int T, int M, float gain, void *stream);
int llzs_fir_mfma_i16_fits(int T, int M);
/* int16 in and out, BIT-EXACT with the reference's double loop (llz_resample.c:583-603, L = 1), screened on the matrix
 * cores (fir_mfma_i8.hip): digits = [5][T] balanced base-256 digits of G[k] = round(g[k] 2^shift), bias = 128 sum G[k],
 * gd = the T double taps, eps = the host's bound on |screen value - reference value| (< 0.25) */
int llzs_fir_mfma_i16x(const short *in, short *out, const short *hist, const signed char *digits, const double *gd,
                       int channels, long n_in, long n_out, long in_pitch, long out_pitch, int T, int M, int shift,
                       long long bias, double gain, double eps, void *stream);
int llzs_fir_mfma_i16x_fits(int T, int M);
#define LLZS_MX_PLANES 5
/* hist_new[c][:] = last (flt_len-1) samples of concat(hist_old[c], in[c][0:n]) */
int llzs_fir_tail_f32(const float *in, const float *hist_old, float *hist_new,
                      int channels, int n, long in_pitch, int flt_len, void *stream);
/* single channel, double, the reference's exact accumulation order (ascending k, multiply then add) */
int llzs_fir_td_f64(const double *in, double *out, const double *hist, const double *taps,
                    int n, int flt_len, void *stream);

/* ---- IIR ---- */
/* cascade of `stages` biquads; coef: stages x 5 doubles {b0,b1,b2,a1,a2}; state: [channels][stages][4] doubles
 * {x1,x2,y1,y2} per stage, read and written. */
int llzs_iir_cascade_f32(const float *in, float *out, const double *coef, double *state,
                         int channels, int n, long in_pitch, long out_pitch, int stages, void *stream);
/* fast path (stage pipeline + lanes along time): n % 1024 == 0, 16-byte aligned rows.  pd = [stages][6][4] powers
 * P^(2^d) of P = A^16 with A = [[-a1,-a2],[1,0]], pl = [stages][64][12] = P^lane, P^(lane%16+1), P^(lane%32+1); same coef / state layout as above. */
#define LLZS_IIR_PIPE_CHUNK 1024     /* 64 lanes x 16 samples */
int llzs_iir_cascade_pipe_f32(const float *in, float *out, const double *coef, const double *pd, const double *pl,
                              const double *state_in, double *state_out /* a different buffer: segments of one launch
                                                                         * are not ordered */, int channels, int n, long in_pitch, long out_pitch, int stages,
                              int warm_chunks /* 0: never split a channel along time */,
                              int float32 /* 1: float32 arithmetic (every section passed the host's noise-gain check) */,
                              void *stream);
/* general direct form I, one channel, double, the reference's exact operation order (llz_iir.c:103-132).
 * xs: N+1 doubles, ys: M+1 doubles (delay lines, read and written) */
/* float32 cascades with a short memory: a wave owns a (channel, time segment) and runs all sections in registers.
 * coef32: [S][5], pd32: [S][16] = P^(2^d) d<4, pl32: [S][64][12], all float; state as above (double). */
int llzs_iir_cascade_wave_f32(const float *in, float *out, const float *coef32, const float *pd32, const float *pl32,
                              const float *ph32 /* [S][24]: (h1[k], h2[k]) k < 8, the outputs at k of the unit start
                                                   * states, then b0 b1 b2 a1 a2 and 3 pad; NULL = the
                                                   * unpacked kernel */,
                              const double *state_in, double *state_out, int channels, int n, long in_pitch, long out_pitch, int stages,
                              int warm_chunks, void *stream);
/* the same with 32 samples per lane and the b0 gains folded into in_gain (iir.hip: k_iir_cascade_wave_pk32): n % 2048 == 0;
 * tables for P = A^32; ph32 [S][40] = (h1[k], h2[k]) k < 16, then 1 b1/b0 b2/b0 a1 a2 xfac yfac pad */
int llzs_iir_cascade_wave32_f32(const float *in, float *out, const float *pd32, const float *pl32, const float *ph32,
                                const double *state_in, double *state_out, int channels, int n, long in_pitch,
                                long out_pitch, int stages, int warm_chunks, float in_gain, void *stream);
/* the same in double for cascades float32 arithmetic is not good enough for (k_iir_cascade_wave_pf64w): cw [S][8] = b1/b0,
 * b2/b0, a1, a2, xfac, yfac, 0, 0; pd [S][16]; plc [S][448] (powers of A^32, see the kernel) */
int llzs_iir_cascade_wave32_f64(const float *in, float *out, const double *cw, const double *pd, const double *plc,
                                const double *state_in, double *state, int channels, int n, long in_pitch, long out_pitch,
                                int stages, int warm_chunks, double in_gain, void *stream);
/* the same in double from the pipelined kernel's tables; at most 8 sections */
int llzs_iir_cascade_wave_f64(const float *in, float *out, const double *coef, const double *pd, const double *pl,
                              const double *state_in, double *state_out, int channels, int n, long in_pitch, long out_pitch, int stages,
                              int warm_chunks, void *stream);
int llzs_iir_df1_f64(const double *in, double *out, const double *a, const double *b, double *xs, double *ys,
                     int M, int N, int n, void *stream);
/* the same recurrence for many channels (iir_df1.hip): float32 in / out, double arithmetic in the reference's order; ab =
 * a[0..ord] then b[0..ord] zero padded (ord = llzs_iir_df1_mc_max_order()); state [channels][2][ord + 1] = the reference's x[]
 * then y[] delay lines, read from state_in and written to state_out (two buffers); segs time segments per channel, later ones
 * warmed up over `warm` samples from zero delay lines */
int llzs_iir_df1_mc_f32(const float *in, float *out, const double *ab, const double *state_in, double *state_out, int channels,
                        long n, long in_pitch, long out_pitch, int M, int N, int segs, int warm, void *stream);
int llzs_iir_df1_mc_max_order(void);

/* ---- resample ---- */
/* y[c][i] = gain * sum_{k<Q} x[c][(i0+i)*M/L - k - in0] * g[(i0+i)%L][k], x before the call start comes from
 * hist[c][Q-1 + idx] (hist holds the previous Q-1 samples).  i0 = global index of the first output of this call,
 * in0 = global index of the first input sample of this call.  n_out outputs per channel. */
int llzs_resample_f32(const float *in, float *out, const float *hist, const float *g,
                      int channels, long n_in, long n_out, long in_pitch, long out_pitch,
                      int L, int M, int Q, float gain, long long i0, long long in0, void *stream);
/* general L/M float32 on the fp32 matrix cores (resample_mfma.hip): atab [ceil(L/16)][steps][64] = the banded tap matrix in
 * MFMA operand order with the gain folded in (steps = llzs_resample_mfma_f32_table_steps), c0tab [ceil(L/16)] =
 * floor(16 t M / L).  The call must start on a period boundary (input index % M == 0, output index % L == 0). */
int llzs_resample_mfma_f32(const float *in, float *out, const float *hist, const float *atab, const int *c0tab,
                           int channels, long n_in, long n_out, long in_pitch, long out_pitch, int L, int M, int Q,
                           void *stream);
int llzs_resample_mfma_f32_fits(int L, int M, int Q);
int llzs_resample_mfma_f32_table_steps(int L, int M, int Q);
/* L = 1 (decimate by M) float32 fast path: gp = M x tp phase taps gp[m][j] = g[j*M+m], zero padded, tp % 16 == 0;
 * input index of output i is i*M (calls start on a period boundary); hist as above. */
int llzs_resample_dec_f32(const float *in, float *out, const float *hist, const float *gp, int channels,
                          long n_in, long n_out, long in_pitch, long out_pitch, int M, int Q, int tp,
                          float gain, void *stream);
int llzs_resample_dec_f32_fits(int M, int tp);     /* 1 when the fast path's LDS image fits */
/* int16 PCM, double taps, double accumulate in ascending k with separate multiply and add, clamp, truncate */
int llzs_resample_i16(const short *in, short *out, const short *hist, const double *g,
                      int channels, long n_in, long n_out, long in_pitch, long out_pitch,
                      int L, int M, int Q, double gain, long long i0, long long in0, void *stream);
/* the same bit-exact int16 result for L >= 2, screened on the int8 matrix cores per phase (resample_i8.hip): atab [ceil(L/16)]
 * [steps][5][64][16] tap digits in operand order (steps = llzs_resample_i16x_ksteps), aoff [ceil(L/16)] band starts, bqtab
 * [16 ceil(L/16)][4] = floor(128 sum_k G_f[k] / 256) as (lo, hi), the phase's own e32, first | last << 8 non-zero tap | exact
 * << 16; g the L x Q double taps, eps the largest per-phase bound, any_exact: some phase carries the exact flag.
 * The call must start on a period boundary (input index % M == 0, output index % L == 0). */
int llzs_resample_i16x(const short *in, short *out, const short *hist, const signed char *atab, const int *aoff,
                       const int *bqtab, const double *g, int channels, long n_in, long n_out, long in_pitch, long out_pitch,
                       int L, int M, int Q, int shift, double gain, double eps, int any_exact, void *stream);
int llzs_resample_i16x_fits(int L, int M, int Q);
int llzs_resample_i16x_ksteps(int L, int M, int Q);
/* the launch a call would make (measurement / documentation): plan[0..6] = waves per workgroup, periods per span, spans per
 * workgroup, workgroups, workgroups resident per CU, LDS bytes per workgroup, 1 when a wave takes several phase tiles */
int llzs_resample_i16x_plan(int L, int M, int Q, int channels, long n_out, int shift, int *plan);
int llzs_tail_i16(const short *in, const short *hist_old, short *hist_new, int channels, long n, long in_pitch,
                  int keep, void *stream);
/* llz_decimate (forward indexed polyphase sum over a history of n samples) and llz_interp (no history),
 * single channel int16, exact order: see llz_resample.c:457-483 and :515-536 */
int llzs_decimate_i16(const short *buf /* n hist + num_in */, short *out, const double *p, int M, int K, int n,
                      int num_out, double gain, void *stream);
int llzs_interp_i16(const short *x /* num_in + K zero padded */, short *out, const double *p, int L, int K,
                    int num_in, double gain, void *stream);

/* ---- FFT ---- */
/* float32 batch, in place; cs = size cos then size sin values (float). inverse: 0 forward, 1 inverse (divides by size) */
int llzs_fft_f32(float *data, int count, int size, const float *cs, int inverse, void *stream);
/* double, one transform, exact reference butterfly order and rounding (no contraction) */
int llzs_fft_f64(double *data, int size, const double *cs, int inverse, void *stream);
/* int32 data, Q15 twiddles (size cos then size sin shorts), bit-exact */
int llzs_fft_fixed(int *data, int count, int size, const short *cs, int inverse, void *stream);

/* ---- correlation ---- */
/* r[k] = sum_i x[i]*y[i+k], k = 0..p, one channel, double, the reference's summation order (llz_corr.c:38-58) */
int llzs_corr_exact_f64(const double *x, const double *y, int n, int p, double *r, void *stream);
/* frames x n float32 -> frames x (p+1), direct form */
int llzs_autocorr_mc_f32(const float *x, float *r, int frames, int n, int p, void *stream);
/* pointwise steps of the FFT form around llzs_fft_f32: real -> zero-padded complex; |X|^2 of the first n bins; 2*Re */
int llzs_acf_pack(const float *x, float *z, int frames, int n, int F, void *stream);
int llzs_acf_power(float *z, int frames, int n, int F, void *stream);
int llzs_acf_extract(const float *z, float *r, int frames, int p, int F, void *stream);
/* the same five steps fused in LDS (fft.hip): one read of the frames, p+1 floats written per frame */
int llzs_acf_fused_f32(const float *x, float *r, int frames, int n, int p, int size, const float *cs, void *stream);

/* windowed-FFT frames (llz_asmodel.c:180-310), float32 batch: x planar [C][frames*F] (row pitch x_pitch), spectra
 * [C][frames][size/2+1]; hist / ola: [C][size-F] state carried between calls (ola_old != ola_new); w: size floats;
 * cs: cos then sin of 2*pi*i/size */
int llzs_stft_analysis_f32(const float *x, const float *hist, float *re, float *im, const float *w, const float *cs,
                           int channels, int frames, int F, int size, long x_pitch, void *stream);
int llzs_stft_synthesis_f32(const float *re, const float *im, float *x, const float *ola_old, float *ola_new,
                            const float *w, const float *cs, int channels, int frames, int F, int size, long x_pitch,
                            float magic, void *stream);

/* MDCT (llz_mdct.c): y[r] = sum_c x[c]*A[r][c] in ascending c, separately rounded multiply and add (device doubles) */
int llzs_matvec_exact_f64(const double *A, const double *x, double *y, int rows, int cols, void *stream);
/* twiddle steps of the FFT forms in double, exact order, device data (mdct.hip): quarter 0 = N-point form, 1 = N/4-point
 * form; post 0 = the step in front of the transform, 1 = the step behind it; cs2: (cos, sin) pairs */
int llzs_mdct_rot_f64(int quarter, int post, const double *in, double *out, const double *cs2, int N, int inverse,
                      double cof, void *stream);
/* fixed-point MDCT steps (mdct_q15.hip), `count` frames per launch, int32 data, Q15 tables, wrapping adds:
 * sums: y[f][r] = sum_c q15(x[f][c], A[r][c]) (llz_mdct_fixed.c:116-152), then (4 y) / quarter_over_n when that is not 0;
 * step: the twiddle steps of the two FFT forms (llz_mdct_fixed.c:155-283); quarter 0 = N-point form, 1 = N/4-point form;
 * post 0 = the step in front of the transform, 1 = the step behind it; tw: (cos, sin) Q15 pairs; cof = Q15 1/sqrt(N) */
int llzs_mdctq_sums(const short *A, const int *x, int *y, int count, int rows, int cols, int quarter_over_n, void *stream);
int llzs_mdctq_step(int quarter, int post, const int *in, int *out, const short *tw, int count, int N, int inverse,
                    int cof, void *stream);
/* the N/4-point form whole in one launch (fold, transform, unfold in LDS): N a power of two in 8..16384; pre / post: the two
 * step tables of this direction; cs: the N/4-point transform's table */
int llzs_mdct4_q15(const int *in, int *out, int count, int N, const short *pre, const short *post, const short *cs,
                   int inverse, int cof, void *stream);
/* the N-point form whole in one launch: N a power of two in 4..4096; pre N pairs, post N/2 (forward) or N (inverse) pairs */
int llzs_mdct1_q15(const int *in, int *out, int count, int N, const short *pre, const short *post, const short *cs, int inverse,
                   void *stream);
/* framing of the single-channel analysis / synthesis symbols in double, exact order (frames_f64.hip; llz_asmodel.c:177-462):
 * slide_window: held_next = (held << hop) ++ fresh, dst = held_next * window (interleaved complex with zero imaginary part
 * when as_complex); split / mirror: interleaved spectrum <-> planes [re 0..N/2 | im 0..N/2]; overlap_add: acc + src * window,
 * the first hop sums leave scaled, the rest slides down into acc_next, zeros enter at the top (acc != acc_next) */
int llzs_frame_slide_window_f64(const double *fresh, const double *held, double *held_next, const double *window,
                                double *dst, int N, int hop, int as_complex, void *stream);
int llzs_spectrum_split_f64(const double *z, double *planes, int bins, void *stream);
int llzs_spectrum_mirror_f64(const double *planes, double *z, int N, void *stream);
int llzs_frame_overlap_add_f64(const double *src, int src_stride, const double *window, const double *acc,
                               double *acc_next, double *leaving, int N, int hop, double scale, void *stream);
int llzs_scale_4_over_n_f64(double *v, int n, double divisor, void *stream);   /* v = (v * 4) / divisor, two roundings */
/* N/4-point-FFT MDCT / IMDCT of `count` float32 frames: forward [count][N] -> [count][N/2], inverse the other way;
 * tc/ts: cos/sin of -2*pi*(k+1/8)/N, k < N/4; cs: cos then sin of 2*pi*i/(N/4) */
int llzs_mdct4_f32(const float *in, float *out, int count, int N, const float *tc, const float *ts, const float *cs,
                   int inverse, void *stream);
/* the same on the register transforms for N in {256, 512, 1024, 2048, 4096, 8192}; LLZ_ERR_RANGE for other N */
int llzs_mdct4_reg_f32(const float *in, float *out, int count, int N, const float *tc, const float *ts, const float *cs,
                       int inverse, void *stream);
/* windowed 50 %-overlap MDCT frames in batch on the same kernels (N in {256 .. 8192}, F = N/2): analysis x [channels][frames F]
 * -> X [channels][frames][F]; synthesis the other way with overlap-add.  win [N]; state_in / state_out [channels][F]
 * (analysis: the previous frame; synthesis: the overlap-add tail), two different buffers */
int llzs_mdct4_frames_f32(const float *in, float *out, int channels, int frames, int N, const float *tc, const float *ts,
                          const float *cs, const float *win, const float *state_in, float *state_out, int inverse,
                          void *stream);

/* ---- PCM ingest / egress ---- */
int llzs_pcm_deinterleave_i16_f32(const short *in, float *out, int channels, long n, float scale, void *stream);
int llzs_pcm_interleave_f32_i16(const float *in, short *out, int channels, long n, float scale, void *stream);

/* ---- synthetic PCM ---- */
int llzs_synth_f32(float *dst, int channels, long n, long pitch, unsigned seed, int chan0, void *stream);
int llzs_synth_i16(short *dst, int channels, long n, long pitch, unsigned seed, int chan0, void *stream);

#ifdef __cplusplus
}
#endif
#endif
