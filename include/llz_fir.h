/*
 * include/llz_fir.h -- FIR design + streaming FIR, C ABI of libllzfilter_hip.so.
 *
 * Part 1 keeps the reference's single-channel `double` API symbol for symbol
 * (reference libllzfilter/llz_fir.h:26-95) so existing callers relink unchanged; the process functions run
 * on the GPU (HIP kernel, same accumulation order, no FMA contraction: results are bit-identical to the
 * reference CPU path for identical taps).
 * Part 2 is the multi-channel float32 batch extension of the same init / process / flush / uninit shape
 *  -- the path the MI355X kernels are built for.
 */
#ifndef LLZ_FIR_H
#define LLZ_FIR_H

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef int win_t;                                   /* reference llz_fir.h:26 */
enum { HAMMING = 0, BLACKMAN, KAISER };              /* reference llz_fir.h:28-32 */

/* ---- Part 1: reference-identical symbols ------------------------------------------------------- */

/* replaces llz_fir.h:38-48 (llz_fir.c:442-530). Handle = pointer cast to unsigned long, as in the reference. */
unsigned long llz_fir_filter_lpf_init(int frame_len, int flt_len, double fc, win_t win_type);
unsigned long llz_fir_filter_hpf_init(int frame_len, int flt_len, double fc, win_t win_type);
unsigned long llz_fir_filter_bandpass_init(int frame_len, int flt_len, double fc1, double fc2, win_t win_type);
unsigned long llz_fir_filter_bandstop_init(int frame_len, int flt_len, double fc1, double fc2, win_t win_type);
void          llz_fir_filter_uninit(unsigned long handle);                       /* llz_fir.h:50 */

/* replaces llz_fir.h:56 (llz_fir.c:547-584). buf_in/buf_out are HOST pointers. Returns frame_len.
 * frame_len must equal the init frame_len: the reference's own history shift is only correct in that case
 * (llz_fir.c:562-566); a different length returns -1 where the reference asserts or
 * silently corrupts its history. */
int llz_fir_filter(unsigned long handle, double *buf_in, double *buf_out, int frame_len);
/* replaces llz_fir.h:58 (llz_fir.c:590-625): emits the flt_len-1 tail samples, returns flt_len-1 */
int llz_fir_filter_flush(unsigned long handle, double *buf_out);

/* windows and estimators, host side (llz_fir.h:64-79, llz_fir.c:61-193) */
int    llz_hamming(double *w, const int N);
int    llz_blackman(double *w, const int N);
int    llz_kaiser(double *w, const int N);
int    llz_kaiser_beta(double *w, const int N, const double beta);
double llz_kaiser_atten2beta(double atten);
int    llz_hamming_cof_num(double ftrans);
int    llz_blackman_cof_num(double ftrans);
int    llz_kaiser_cof_num(double ftrans, double atten);

/* tap design, host side (llz_fir.h:86-92, llz_fir.c:271-393). *h is malloc'ed; the CALLER frees it. */
int llz_fir_lpf_cof(double **h, int N, double fc, win_t win_type);
int llz_fir_hpf_cof(double **h, int N, double fc, win_t win_type);
int llz_fir_bandpass_cof(double **h, int N, double fc1, double fc2, win_t win_type);
int llz_fir_bandstop_cof(double **h, int N, double fc1, double fc2, win_t win_type);

/* llz_fir.h:94 (llz_fir.c:411-426): host dot product, x points at the newest sample */
double llz_conv(const double *x, const double *h, int h_len);

/* ---- Part 2: multi-channel float32 batch extension ---------------------------------------------- */

enum {
    LLZ_FIR_ALGO_AUTO = 0,       /* time domain up to 32 taps, overlap-save for 33..257 (1024-point), 258..513
                                  * (2048-point), 514..1025 (4096-point) and 1026..6145 (8192-point), matrix-core time
                                  * domain beyond */
    LLZ_FIR_ALGO_TIME = 1,       /* direct form, taps broadcast, input window staged in LDS */
    LLZ_FIR_ALGO_OVERLAP_SAVE = 2, /* 1024-point in-LDS FFT overlap-save, flt_len <= 257 */
    LLZ_FIR_ALGO_TIME_MFMA = 3,   /* direct form as a banded Toeplitz product on the fp32 matrix cores */
    LLZ_FIR_ALGO_OVERLAP_SAVE_2048 = 4, /* 2048-point register-transform overlap-save, 2 <= flt_len <= 1025 */
    LLZ_FIR_ALGO_OVERLAP_SAVE_4096 = 5, /* 4096-point register-transform overlap-save, 2 <= flt_len <= 3073 */
    LLZ_FIR_ALGO_OVERLAP_SAVE_8192 = 6  /* 8192-point register-transform overlap-save on pairs of waves, 2 <= flt_len <= 6145 */
};

/* channels independent filters sharing one tap set. taps: HOST pointer, flt_len floats (double variant
 * below rounds to float once).  Returns (unsigned long)-1 on failure (llz_hip_last_error() says why). */
unsigned long llz_fir_filter_mc_init(int channels, int frame_len, const float *taps, int flt_len, int algo);
unsigned long llz_fir_filter_mc_init_f64taps(int channels, int frame_len, const double *taps, int flt_len, int algo);
/* design + init in one call, mirroring llz_fir_filter_{lpf,hpf,bandpass,bandstop}_init */
unsigned long llz_fir_filter_mc_lpf_init(int channels, int frame_len, int flt_len, double fc, win_t win_type);
unsigned long llz_fir_filter_mc_hpf_init(int channels, int frame_len, int flt_len, double fc, win_t win_type);
unsigned long llz_fir_filter_mc_bandpass_init(int channels, int frame_len, int flt_len, double fc1, double fc2, win_t win_type);
unsigned long llz_fir_filter_mc_bandstop_init(int channels, int frame_len, int flt_len, double fc1, double fc2, win_t win_type);
void          llz_fir_filter_mc_uninit(unsigned long handle);

/* planar [channels][frame_len] float32 in and out (out may not alias in). Pointers may be device memory
 * (used in place, asynchronous on the handle's stream) or host memory (staged through the GPU, synchronous).
 * frame_len must equal the init frame_len. Returns frame_len, or a negative LLZ_ERR_* code. */
int llz_fir_filter_mc(unsigned long handle, const float *in, float *out, int frame_len);
/* out: planar [channels][flt_len-1]; returns flt_len-1 */
int llz_fir_filter_mc_flush(unsigned long handle, float *out);
int llz_fir_filter_mc_flt_len(unsigned long handle);
int llz_fir_filter_mc_algo(unsigned long handle);
/* stream: a hipStream_t passed as void* (NULL = default stream) */
int llz_fir_filter_mc_set_stream(unsigned long handle, void *stream);

#ifdef __cplusplus
}
#endif
#endif
