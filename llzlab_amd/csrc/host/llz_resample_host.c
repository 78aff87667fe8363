/*
 * llz_resample_host.c -- handle layer of the resample family: the reference's int16 single-channel symbols
 * (reference libllzfilter/llz_resample.c:124-617) and the multi-channel batch extension.  Prototype design and
 * the polyphase / time-varying tap matrices are host C (setup time); every sample is computed on the device.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../../../include/llz_resample.h"
#include "llz_host.h"

/* ---- tap matrices (llz_resample.c:124-176 polyphase, :193-255 time-varying) ---- */

typedef struct {
    int n;          /* prototype length */
    int rows, cols; /* phases x taps per phase */
    double *h;      /* prototype low-pass */
    double *mat;    /* rows x cols, row major */
} tapmat_t;

static void tapmat_free(tapmat_t *t)
{
    free(t->h); free(t->mat);
    t->h = t->mat = NULL;
}

static int proto_estimate(double ftrans, win_t win)
{
    switch (win) {
    case HAMMING:  return llz_hamming_cof_num(ftrans);
    case BLACKMAN: return llz_blackman_cof_num(ftrans);
    case KAISER:   return llz_kaiser_cof_num(ftrans, 90);     /* llz_resample.c:143 / :212 */
    }
    return -1;
}

/* phases = M (decimate), L (interp / rational).  tv_M = 0 selects the polyphase fill p[i][j] = g*h[phases*j + i];
 * tv_M > 0 selects the time-varying fill g[i][j] = g*h[j*phases + (i*tv_M) % phases]. */
static int tapmat_build(tapmat_t *t, int phases, int tv_M, double fc, double mgain, win_t win)
{
    if (mgain == 0) mgain = 1.0;                               /* llz_resample.c:131-132 */
    const int est = proto_estimate(0.15 * fc, win);            /* transition band = 0.15 * fc */
    if (est < 0) {
        llzs_set_error("resample init: unknown window %d", win);
        return LLZ_ERR_ARG;
    }
    t->n = 2 * (est / (2 * phases)) * phases + 1;              /* odd prototype: :151-152 / :218-219 */
    t->rows = phases;
    t->cols = t->n / phases + 1;
    if (llz_host_design(LLZ_KIND_LPF, &t->h, t->n, fc, 0.0, win) != t->n) return LLZ_ERR_ARG;
    t->mat = (double *)calloc((size_t)t->rows * t->cols, sizeof(double));
    if (!t->mat) return LLZ_ERR_NOMEM;
    for (int i = 0; i < t->rows; i++)
        for (int j = 0; j < t->cols; j++) {
            const int u = tv_M ? j * phases + (i * tv_M) % phases : phases * j + i;
            if (u < t->n) t->mat[(size_t)i * t->cols + j] = mgain * t->h[u];
        }
    return LLZ_OK;
}

static int gcd_int(int a, int b)
{
    while (b) { const int r = a % b; a = b; b = r; }
    return a;
}

/* =====================================================================================================
 * Part 1: reference-identical int16 single-channel API
 * ===================================================================================================== */

enum { RS_DECIMATE = 0, RS_INTERP = 1, RS_RATIONAL = 2 };

typedef struct {
    int tag;
    int mode, L, M;
    double gain;
    tapmat_t taps;
    int num_in, num_out;
    long long out_index;        /* running output count: phase = out_index % L (llz_resample.c:586) */
    int hist;                   /* samples kept in front of each frame */
    short *buf;                 /* host: [hist | num_in | cols zero slack] */
    short *d_buf, *d_out;       /* device mirrors */
    double *d_mat;
} rs1_t;

static void rs1_destroy(rs1_t *r)
{
    if (!r) return;
    tapmat_free(&r->taps);
    free(r->buf);
    llzs_free(r->d_buf); llzs_free(r->d_out); llzs_free(r->d_mat);
    r->tag = 0;
    free(r);
}

static unsigned long rs1_finish(rs1_t *r)
{
    const size_t span = (size_t)r->hist + r->num_in + r->taps.cols + 1;
    const size_t mat_bytes = sizeof(double) * (size_t)r->taps.rows * r->taps.cols;
    r->buf = (short *)calloc(span, sizeof(short));             /* zero history: llz_resample.c:299-300 */
    r->d_buf = (short *)llzs_malloc(span * sizeof(short));
    r->d_out = (short *)llzs_malloc(sizeof(short) * (size_t)r->num_out);
    r->d_mat = (double *)llzs_malloc(mat_bytes);
    if (!r->buf || !r->d_buf || !r->d_out || !r->d_mat ||
        llzs_h2d(r->d_mat, r->taps.mat, mat_bytes, NULL) != LLZ_OK) {
        rs1_destroy(r);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)r;
}

static rs1_t *rs1_new(int mode, int L, int M, double gain)
{
    rs1_t *r = (rs1_t *)calloc(1, sizeof(*r));
    if (!r) return NULL;
    r->tag = LLZ_TAG_RS1;
    r->mode = mode; r->L = L; r->M = M; r->gain = gain;
    return r;
}

unsigned long llz_decimate_init(int M, double gain, win_t win_type)
{
    if (M < 1 || M > LLZ_RATIO_MAX) {                          /* llz_resample.c:278-279 */
        llzs_set_error("llz_decimate_init: M=%d outside 1..%d", M, LLZ_RATIO_MAX);
        return LLZ_BAD_HANDLE;
    }
    rs1_t *r = rs1_new(RS_DECIMATE, 1, M, gain);
    if (!r) return LLZ_BAD_HANDLE;
    if (tapmat_build(&r->taps, M, 0, 1. / M, 1, win_type) != LLZ_OK) { rs1_destroy(r); return LLZ_BAD_HANDLE; }
    r->num_out = LLZ_DEFAULT_FRAMELEN / M;                     /* :291-293 */
    r->num_in = r->num_out * M;
    r->hist = r->taps.n;                                       /* :297: history of n samples */
    return rs1_finish(r);
}

unsigned long llz_interp_init(int L, double gain, win_t win_type)
{
    if (L < 1 || L > LLZ_RATIO_MAX) {                          /* :326-327 */
        llzs_set_error("llz_interp_init: L=%d outside 1..%d", L, LLZ_RATIO_MAX);
        return LLZ_BAD_HANDLE;
    }
    rs1_t *r = rs1_new(RS_INTERP, L, 1, gain);
    if (!r) return LLZ_BAD_HANDLE;
    if (tapmat_build(&r->taps, L, 0, 1. / L, L, win_type) != LLZ_OK) { rs1_destroy(r); return LLZ_BAD_HANDLE; }
    r->num_in = LLZ_DEFAULT_FRAMELEN;                          /* :338-339 */
    r->num_out = LLZ_DEFAULT_FRAMELEN * L;
    r->hist = 0;
    return rs1_finish(r);
}

unsigned long llz_resample_filter_init(int L, int M, double gain, win_t win_type)
{
    if (L < 1 || M < 1) {
        llzs_set_error("llz_resample_filter_init: L=%d M=%d", L, M);
        return LLZ_BAD_HANDLE;
    }
    const double ratio = ((double)L) / M;
    if (ratio > LLZ_RATIO_MAX || (1. / ratio) > LLZ_RATIO_MAX) {   /* :375-378 */
        llzs_set_error("llz_resample_filter_init: ratio %d/%d outside 1/%d..%d", L, M, LLZ_RATIO_MAX, LLZ_RATIO_MAX);
        return LLZ_BAD_HANDLE;
    }
    rs1_t *r = rs1_new(RS_RATIONAL, L, M, gain);
    if (!r) return LLZ_BAD_HANDLE;
    const double fc = (1. / L < 1. / M) ? 1. / L : 1. / M;     /* :382 */
    if (tapmat_build(&r->taps, L, M, fc, L, win_type) != LLZ_OK) { rs1_destroy(r); return LLZ_BAD_HANDLE; }
    r->num_in = (L * M) / gcd_int(L, M);                       /* :394-396: lcm doubled up to >= 1024 */
    while (r->num_in < LLZ_DEFAULT_FRAMELEN) r->num_in *= 2;
    r->num_out = (r->num_in * L) / M;
    r->hist = r->taps.cols;                                    /* :402: Q samples kept */
    return rs1_finish(r);
}

static void rs1_uninit(unsigned long handle)
{
    if (LLZ_HANDLE_OK(handle, rs1_t, LLZ_TAG_RS1))
        rs1_destroy((rs1_t *)handle);
}

/* the reference's example calls llz_resample_filter_uninit on every kind of handle (main.c:125);
 * all three release the same structure here, so that call is harmless instead of a crash */
void llz_decimate_uninit(unsigned long handle)        { rs1_uninit(handle); }
void llz_interp_uninit(unsigned long handle)          { rs1_uninit(handle); }
void llz_resample_filter_uninit(unsigned long handle) { rs1_uninit(handle); }

int llz_get_resample_framelen_bytes(unsigned long handle)
{
    if (!LLZ_HANDLE_OK(handle, rs1_t, LLZ_TAG_RS1)) return LLZ_ERR_ARG;
    return 2 * ((rs1_t *)handle)->num_in;
}

static int rs1_process(unsigned long handle, int mode, unsigned char *in, int in_bytes, unsigned char *out,
                       int *out_bytes, const char *who)
{
    if (!LLZ_HANDLE_OK(handle, rs1_t, LLZ_TAG_RS1) || !in || !out || !out_bytes) {
        llzs_set_error("%s: bad handle or NULL buffer", who);
        return LLZ_ERR_ARG;
    }
    rs1_t *r = (rs1_t *)handle;
    if (r->mode != mode) {
        llzs_set_error("%s: handle was created by a different *_init", who);
        return LLZ_ERR_ARG;
    }
    if (in_bytes != 2 * r->num_in) {                           /* reference: assert (:443, :507, :560) */
        llzs_set_error("%s: %d input bytes, expected %d", who, in_bytes, 2 * r->num_in);
        return LLZ_ERR_ARG;
    }
    /* [history | frame]; the history is the tail of the previous buffer (:452-455 / :571-576) */
    memmove(r->buf, r->buf + r->num_in, sizeof(short) * (size_t)r->hist);
    memcpy(r->buf + r->hist, in, (size_t)in_bytes);
    const size_t span = (size_t)r->hist + r->num_in + r->taps.cols;
    int rc = llzs_h2d(r->d_buf, r->buf, span * sizeof(short), NULL);
    if (rc == LLZ_OK) {
        if (mode == RS_DECIMATE)
            rc = llzs_decimate_i16(r->d_buf, r->d_out, r->d_mat, r->M, r->taps.cols, r->taps.n, r->num_out,
                                   r->gain, NULL);
        else if (mode == RS_INTERP)
            rc = llzs_interp_i16(r->d_buf, r->d_out, r->d_mat, r->L, r->taps.cols, r->num_in, r->gain, NULL);
        else
            /* frame-local indexing as the reference: x + (i*M)/L with phase out_index % L; frames hold a whole
             * number of L/M periods, so local and global indexing coincide */
            rc = llzs_resample_i16(r->d_buf + r->hist, r->d_out, r->d_buf + 1, r->d_mat, 1, r->num_in,
                                   r->num_out, r->num_in, r->num_out, r->L, r->M, r->taps.cols, r->gain,
                                   r->out_index % r->L, 0, NULL);
    }
    if (rc == LLZ_OK) rc = llzs_d2h(out, r->d_out, sizeof(short) * (size_t)r->num_out, NULL);
    if (rc != LLZ_OK) return rc;
    r->out_index += r->num_out;
    *out_bytes = 2 * r->num_out;
    return 0;
}

int llz_decimate(unsigned long handle, unsigned char *sample_in, int sample_in_size, unsigned char *sample_out,
                 int *sample_out_size)
{
    return rs1_process(handle, RS_DECIMATE, sample_in, sample_in_size, sample_out, sample_out_size, "llz_decimate");
}

int llz_interp(unsigned long handle, unsigned char *sample_in, int sample_in_size, unsigned char *sample_out,
               int *sample_out_size)
{
    return rs1_process(handle, RS_INTERP, sample_in, sample_in_size, sample_out, sample_out_size, "llz_interp");
}

int llz_resample(unsigned long handle, unsigned char *sample_in, int sample_in_size, unsigned char *sample_out,
                 int *sample_out_size)
{
    return rs1_process(handle, RS_RATIONAL, sample_in, sample_in_size, sample_out, sample_out_size, "llz_resample");
}

/* =====================================================================================================
 * Part 2: multi-channel rational resampler
 * ===================================================================================================== */

typedef struct {
    int tag;
    int channels, L, M, fmt, Q;
    double gain;
    tapmat_t taps;
    void *d_mat;                /* float (F32) or double (I16) L x Q */
    float *d_phase;             /* F32, L == 1: M x tp phase taps for the polyphase fast path (NULL otherwise) */
    int tp;
    int use_mfma;               /* F32, L == 1: decimating FIR on the matrix cores (fir_mfma.hip) */
    /* I16, L == 1: the bit-exact path screened on the matrix cores (fir_mfma_i8.hip) */
    int use_screen, screen_shift;
    long long screen_bias;
    double screen_eps;
    int screen_any_exact;               /* general L/M: some phase is a single tap 1.0 at gain 1.0 (its outputs are integers) */
    signed char *d_digits;      /* [LLZS_MX_PLANES][Q] */
    /* I16, L >= 2: the same screen per phase, on the phase-tile mapping (resample_i8.hip) */
    int use_screen_lm;
    signed char *d_scr_atab;    /* [ceil(L/16)][steps][5][64][16] tap digits in matrix-core operand order */
    int *d_scr_aoff;            /* [ceil(L/16)] band starts */
    int *d_scr_bq;              /* [16 ceil(L/16)][4] per phase: bias (lo, hi), e32, exact flag */
    /* F32, L >= 5: the banded tap matrix in matrix-core operand order (resample_mfma.hip) */
    float *d_band;
    int *d_band_c0;
    double *d_gd;                       /* LLZ_PCM_I16_FAST on the screened kernel: the double taps of its second looks */
    void *d_hist[2];            /* [channels][Q-1] samples of the handle's format, ping-pong */
    int cur;
    long long in_count, out_count;   /* samples consumed / produced per channel so far */
    void *stream;
    int device;                 /* the device the handle's buffers live on: every call binds it */
    llz_stage_t st_in, st_out;
} rsm_t;

static size_t rsm_sample_bytes(const rsm_t *r) { return r->fmt == LLZ_PCM_F32 ? sizeof(float) : sizeof(short); }

static void rsm_destroy(rsm_t *r)
{
    if (!r) return;
    tapmat_free(&r->taps);
    llzs_free(r->d_mat); llzs_free(r->d_phase); llzs_free(r->d_digits); llzs_free(r->d_scr_atab); llzs_free(r->d_scr_aoff);
    llzs_free(r->d_scr_bq); llzs_free(r->d_band); llzs_free(r->d_band_c0); llzs_free(r->d_gd); llzs_free(r->d_hist[0]); llzs_free(r->d_hist[1]);
    llz_stage_release(&r->st_in); llz_stage_release(&r->st_out);
    r->tag = 0;
    free(r);
}

/* Screen tables of the bit-exact int16 decimator (fir_mfma_i8.hip): fixed-point taps G[k] = round(gain g[k] 2^shift) as
 * five balanced base-256 digits, the constant 128 sum G[k] of the samples' +128 offset, and eps, a bound on the distance
 * between the screen's value v = S 2^-shift and the reference's double result y = fl(fl(sum x g) gain)
 * (llz_resample.c:590-594), for |x| <= 32768 and u = 2^-53:
 *   tap quantisation     |sum x (gain g - G 2^-shift)|    <= 32768 sum_k |fl(gain g[k]) - G[k] 2^-shift|   (evaluated below)
 *                        + the rounding of fl(gain g[k])   <= 32768 u |gain| sum_k |g[k]|
 *   reference's sum      |fl(sum x g) - sum x g|          <= Q u / (1 - Q u) sum |x g| <= Q 2^-52 32768 sum_k |g[k]|
 *   reference's * gain   one more rounding of a value <= 32768 |gain| sum|g|
 * eps = 2 (the sum of those) + 2^-30 + (the integer terms the kernel does not form, see below): the factor 2 and the constant
 * are margin (they also cover this function's own floating-point sums).  Returns 0 when the taps do not suit the screen (the all-double kernel then runs). */
static int rsm_build_screen(rsm_t *r)
{
    const int Q = r->Q;
    const double *g = r->taps.mat;
    const double gain = r->gain;
    double maxabs = 0.0, sumabs = 0.0;
    for (int k = 0; k < Q; k++) {
        const double a = fabs(g[k] * gain);
        if (!(a < 1e30)) return 0;
        if (a > maxabs) maxabs = a;
        sumabs += a;
    }
    if (maxabs == 0.0 || !llzs_fir_mfma_i16x_fits(Q, r->M)) return 0;
    int e_max, e_sum;
    (void)frexp(maxabs, &e_max);                        /* maxabs < 2^e_max */
    (void)frexp(sumabs, &e_sum);
    int shift = 38 - e_max;                             /* |G| < 2^38: five balanced digits hold it */
    if (shift > 46 - e_sum) shift = 46 - e_sum;         /* 2^15 sum|G| < 2^62: the int64 total cannot wrap */
    if (shift > 46) shift = 46;
    if (shift < 32) return 0;                           /* (gains above ~64: the integer decision needs shift >= 32) */
    signed char *digits = (signed char *)malloc((size_t)LLZS_MX_PLANES * Q);
    if (!digits) return 0;
    long long sumG = 0, sum_d0 = 0;
    double qerr = 0.0;
    int ok = 1;
    for (int k = 0; k < Q; k++) {
        const double gk = g[k] * gain;
        long long G = llround(ldexp(gk, shift));
        qerr += fabs(gk - ldexp((double)G, -shift));
        sumG += G;
        for (int p = 0; p < LLZS_MX_PLANES; p++) {
            const int d = (int)(((G + 128) & 255) - 128);            /* balanced digit in [-128, 127] */
            digits[(size_t)p * Q + k] = (signed char)d;
            if (p == 0) sum_d0 += d < 0 ? -d : d;
            G = (G - d) / 256;
        }
        if (G != 0) ok = 0;
    }
    /* the kernel leaves out two integer terms of S = sum x G: the product of the samples' low digit (|.| <= 128) with the
     * taps' lowest digit, |.| <= 128 sum|d_0|, and the low 8 bits of the bias 128 sum G, < 256 -- both exact bounds, in
     * units of 2^-shift */
    const double eps = 2.0 * 32768.0 * (qerr + (double)(Q + 2) * ldexp(1.0, -52) * sumabs) + ldexp(1.0, -30) +
                       ldexp(128.0 * (double)sum_d0 + 256.0, -shift);
    if (ok && eps < 0.0625) {
        if (!r->d_digits) r->d_digits = (signed char *)llzs_malloc((size_t)LLZS_MX_PLANES * Q);
        ok = r->d_digits && llzs_h2d_table(r->d_digits, digits, (size_t)LLZS_MX_PLANES * Q) == LLZ_OK;
    } else {
        ok = 0;
    }
    free(digits);
    if (ok) {
        r->screen_shift = shift;
        r->screen_bias = 128 * sumG;
        r->screen_eps = eps;
    }
    return ok;
}

/* The same screen for L >= 2 (resample_i8.hip): one tap row per phase f, g_f[k] = taps.mat[f][k]; ONE shift for all rows (from
 * the largest |gain g|), per-row digits, bias and error bound (the formulas of rsm_build_screen with the row's own sums), eps =
 * the largest row bound.  The digits go straight into matrix-core operand order: phase tile t = phases 16 t .. 16 t + 15; its
 * band starts at window position a_t = c_{16t} + Hp - (Q-1) (Hp = Q-1 rounded up to 8, the position of a period's own
 * first sample; c_f = (f M) / L); lane (r = lane % 16, kq = lane / 16) of step s and digit plane p holds, in byte j, digit p of
 * the tap of phase f = 16 t + r that multiplies window position pos = a_t + 64 s + 16 chunk(kq) + j, i.e. tap k = c_f + Hp - pos
 * (chunk(kq) = ((kq & 1) << 1) | (kq >> 1): kernels/screen_i8.hpp). */
static int rsm_build_screen_lm(rsm_t *r)
{
    const int L = r->L, M = r->M, Q = r->Q, nt = (L + 15) / 16;
    const double gain = r->gain;
    if (!llzs_resample_i16x_fits(L, M, Q)) return 0;
    double maxabs = 0.0, max_sumabs = 0.0;
    for (int f = 0; f < L; f++) {
        double sumabs = 0.0;
        for (int k = 0; k < Q; k++) {
            const double a = fabs(r->taps.mat[(size_t)f * Q + k] * gain);
            if (!(a < 1e30)) return 0;
            if (a > maxabs) maxabs = a;
            sumabs += a;
        }
        if (sumabs > max_sumabs) max_sumabs = sumabs;
    }
    if (maxabs == 0.0) return 0;
    int e_max, e_sum;
    (void)frexp(maxabs, &e_max);
    (void)frexp(max_sumabs, &e_sum);
    int shift = 38 - e_max;                             /* |G| < 2^38: five balanced digits hold it */
    if (shift > 46 - e_sum) shift = 46 - e_sum;         /* 2^15 sum|G| < 2^62 */
    if (shift > 46) shift = 46;
    if (shift < 32) return 0;
    const int steps = llzs_resample_i16x_ksteps(L, M, Q), Hp = (Q - 1 + 7) & ~7;
    const size_t abytes = (size_t)nt * steps * 5 * 1024;
    signed char *atab = (signed char *)calloc(abytes, 1);
    signed char *dig = (signed char *)malloc((size_t)5 * Q);
    int *aoff = (int *)calloc((size_t)nt, sizeof(int)), *bq = (int *)calloc((size_t)nt * 64, sizeof(int));
    int ok = atab && dig && aoff && bq;
    double eps = 0.0;
    int any_exact = 0;
    for (int f = 0; ok && f < L; f++) {
        const double *g = r->taps.mat + (size_t)f * Q;
        const int t = f / 16, row = f % 16, cf = (int)(((long)f * M) / L);
        const int a_t = (int)(((long)16 * t * M) / L) + Hp - (Q - 1);
        aoff[t] = a_t;
        long long sumG = 0, sum_d0 = 0;
        double qerr = 0.0, sumabs = 0.0;
        int nonzero = 0, unit = 0, kfirst = Q - 1, klast = 0;
        for (int k = 0; k < Q; k++) {
            const double gk = g[k] * gain;
            long long G = llround(ldexp(gk, shift));
            qerr += fabs(gk - ldexp((double)G, -shift));
            sumabs += fabs(gk);
            sumG += G;
            nonzero += g[k] != 0.0;
            unit += g[k] == 1.0;
            if (g[k] != 0.0) {
                if (k < kfirst) kfirst = k;
                if (k > klast) klast = k;
            }
            for (int p = 0; p < 5; p++) {
                const int d = (int)(((G + 128) & 255) - 128);
                dig[(size_t)p * Q + k] = (signed char)d;
                if (p == 0) sum_d0 += d < 0 ? -d : d;
                G = (G - d) / 256;
            }
            if (G != 0) ok = 0;
        }
        const double e = 2.0 * 32768.0 * (qerr + (double)(Q + 2) * ldexp(1.0, -52) * sumabs) + ldexp(1.0, -30) +
                         ldexp(128.0 * (double)sum_d0 + 256.0, -shift);
        if (e > eps) eps = e;
        const long long bias = 128 * sumG, bqv = bias >= 0 ? bias / 256 : -((-bias + 255) / 256);      /* floor(bias / 256) */
        bq[4 * f] = (int)(unsigned)((unsigned long long)bqv & 0xffffffffull);
        bq[4 * f + 1] = (int)(bqv >> 32);
        bq[4 * f + 2] = (int)((unsigned)ceil(ldexp(e, 32)) + 2u);
        /* a phase whose only non-zero tap is exactly 1.0, at gain 1.0 (phase 0 whenever fc = 1/L: the windowed sinc is zero at
         * the other multiples of L): the reference's y is the sample itself, an integer, and so is the screen's value -- bit
         * for bit (G = 2^shift, every digit product exact, lowest digit and bias remainder zero).  Every such output would
         * otherwise take the recompute path: one lane row in 16 of that phase tile, 47 double steps each. */
        if (nonzero == 0) kfirst = klast = 0;
        /* (the recompute path runs taps kfirst .. klast only: zero taps in front and behind add +-0 to the reference's sum) */
        /* (a phase without any tap: every output is the integer 0, whatever the gain) */
        const int exact = nonzero == 0 || (nonzero == 1 && unit == 1 && gain == 1.0);
        any_exact |= exact;
        bq[4 * f + 3] = kfirst | (klast << 8) | (exact ? 1 << 16 : 0);
        for (int s = 0; s < steps; s++)
            for (int kq = 0; kq < 4; kq++)
                for (int j = 0; j < 16; j++) {
                    const int pos = a_t + 64 * s + 16 * (((kq & 1) << 1) | (kq >> 1)) + j;
                    const int k = cf + Hp - pos;
                    if (k < 0 || k >= Q) continue;
                    for (int p = 0; p < 5; p++)
                        atab[((((size_t)t * steps + s) * 5 + p) * 64 + (size_t)(16 * kq + row)) * 16 + j] = dig[(size_t)p * Q + k];
                }
    }
    if (ok && eps < 0.0625) {
        if (!r->d_scr_atab) r->d_scr_atab = (signed char *)llzs_malloc(abytes);
        if (!r->d_scr_aoff) r->d_scr_aoff = (int *)llzs_malloc(sizeof(int) * (size_t)nt);
        if (!r->d_scr_bq) r->d_scr_bq = (int *)llzs_malloc(sizeof(int) * (size_t)nt * 64);
        ok = r->d_scr_atab && r->d_scr_aoff && r->d_scr_bq && llzs_h2d_table(r->d_scr_atab, atab, abytes) == LLZ_OK &&
             llzs_h2d_table(r->d_scr_aoff, aoff, sizeof(int) * (size_t)nt) == LLZ_OK &&
             llzs_h2d_table(r->d_scr_bq, bq, sizeof(int) * (size_t)nt * 64) == LLZ_OK;
    } else {
        ok = 0;
    }
    free(atab); free(dig); free(aoff); free(bq);
    if (ok) {
        r->screen_shift = shift;
        r->screen_eps = eps;
        r->screen_any_exact = any_exact;
    }
    return ok;
}

static int rsm_upload_matrix(rsm_t *r)
{
    const size_t count = (size_t)r->L * r->Q;
    if (r->fmt == LLZ_PCM_I16) {
        const int rc = llzs_h2d_table(r->d_mat, r->taps.mat, sizeof(double) * count);
        const int want = rc == LLZ_OK && llzs_tune(LLZS_TUNE_RS_I16_PATH) != 1;
        r->use_screen = want && r->L == 1 && rsm_build_screen(r);
        r->use_screen_lm = want && r->L >= 2 && rsm_build_screen_lm(r);
        return rc;
    }
    float *m32 = (float *)malloc(sizeof(float) * count);
    if (!m32) return LLZ_ERR_NOMEM;
    for (size_t i = 0; i < count; i++) m32[i] = (float)r->taps.mat[i];
    int rc = llzs_h2d_table(r->d_mat, m32, sizeof(float) * count);
    /* LLZ_PCM_I16_FAST promises the reference's result within 1 LSB.  The bit-exact screened kernel is the faster of the two since
     * round 3 (BASELINE config 5: 23 ms against 31 ms for the float32-sum kernel): a FAST handle takes it whenever the screen
     * accepts the taps, and keeps the float32-sum kernel for the rest (gains above ~64, frames shorter than a tile). */
    r->use_screen = 0;
    if (rc == LLZ_OK && r->fmt == LLZ_PCM_I16_FAST && r->L == 1 && llzs_tune(LLZS_TUNE_RS_I16_PATH) != 1) {
        if (!r->d_gd) r->d_gd = (double *)llzs_malloc(sizeof(double) * count);
        r->use_screen = r->d_gd && llzs_h2d_table(r->d_gd, r->taps.mat, sizeof(double) * count) == LLZ_OK && rsm_build_screen(r);
    }
    if (rc == LLZ_OK && r->fmt == LLZ_PCM_F32 && llzs_resample_mfma_f32_fits(r->L, r->M, r->Q) &&
        llzs_tune(LLZS_TUNE_RS_GENERIC) < 1) {
        /* phase tile t = phases 16t .. 16t+15; its band starts at input offset c0 - (Q-1), c0 = floor(16 t M / L), and is
         * walked 4 samples per matrix-core step: lane (r = lane % 16, kq = lane / 16) of step s holds the tap of phase
         * f = 16t + r that multiplies the band's sample 4s + kq, i.e. g_f[c_f - c0 + (Q-1) - (4s + kq)] */
        const int steps = llzs_resample_mfma_f32_table_steps(r->L, r->M, r->Q), nt = (r->L + 15) / 16;
        float *band = (float *)calloc((size_t)nt * steps * 64, sizeof(float));
        int *c0 = (int *)malloc(sizeof(int) * (size_t)nt);
        if (!band || !c0) { free(band); free(c0); free(m32); return LLZ_ERR_NOMEM; }
        for (int t = 0; t < nt; t++) {
            c0[t] = (int)(((long)16 * t * r->M) / r->L);
            for (int s = 0; s < steps; s++)
                for (int lane = 0; lane < 64; lane++) {
                    const int f = 16 * t + (lane & 15), u = 4 * s + (lane >> 4);
                    if (f >= r->L) continue;
                    const int k = (int)(((long)f * r->M) / r->L) - c0[t] + (r->Q - 1) - u;
                    if (k >= 0 && k < r->Q)
                        band[((size_t)t * steps + s) * 64 + lane] = (float)(m32[(size_t)f * r->Q + k] * (float)r->gain);
                }
        }
        if (!r->d_band) r->d_band = (float *)llzs_malloc(sizeof(float) * (size_t)nt * steps * 64);
        if (!r->d_band_c0) r->d_band_c0 = (int *)llzs_malloc(sizeof(int) * (size_t)nt);
        rc = (r->d_band && r->d_band_c0) ? LLZ_OK : LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) rc = llzs_h2d_table(r->d_band, band, sizeof(float) * (size_t)nt * steps * 64);
        if (rc == LLZ_OK) rc = llzs_h2d_table(r->d_band_c0, c0, sizeof(int) * (size_t)nt);
        free(band); free(c0);
    }
    if (rc == LLZ_OK && r->L == 1) {
        r->use_mfma = llzs_fir_mfma_f32_fits(r->Q, r->M) && llzs_tune(LLZS_TUNE_RS_DEC_VALU) != 1;
        /* phase taps for the decimator fast path: gp[m][j] = g[0][j*M + m], rows zero padded to tp */
        const int per_phase = (r->Q + r->M - 1) / r->M;
        r->tp = (per_phase + 15) & ~15;
        if (llzs_resample_dec_f32_fits(r->M, r->tp)) {
            float *gp = (float *)calloc((size_t)r->M * r->tp, sizeof(float));
            if (!gp) { free(m32); return LLZ_ERR_NOMEM; }
            for (int k = 0; k < r->Q; k++) gp[(size_t)(k % r->M) * r->tp + k / r->M] = m32[k];
            if (!r->d_phase) r->d_phase = (float *)llzs_malloc(sizeof(float) * (size_t)r->M * r->tp);
            rc = r->d_phase ? llzs_h2d_table(r->d_phase, gp, sizeof(float) * (size_t)r->M * r->tp) : LLZ_ERR_NOMEM;
            free(gp);
        }
    }
    free(m32);
    return rc;
}

unsigned long llz_resample_mc_init(int channels, int L, int M, double gain, win_t win_type, int pcm_format)
{
    if (channels < 1 || channels > 65535 || L < 1 || M < 1 ||
        (pcm_format != LLZ_PCM_F32 && pcm_format != LLZ_PCM_I16 && pcm_format != LLZ_PCM_I16_FAST)) {
        llzs_set_error("llz_resample_mc_init: channels %d L %d M %d format %d", channels, L, M, pcm_format);
        return LLZ_BAD_HANDLE;
    }
    if (pcm_format == LLZ_PCM_I16_FAST && L != 1) {
        llzs_set_error("llz_resample_mc_init: LLZ_PCM_I16_FAST needs L == 1 (got %d/%d); use LLZ_PCM_I16", L, M);
        return LLZ_BAD_HANDLE;
    }
    const double ratio = ((double)L) / M;
    if (ratio > LLZ_RATIO_MAX || (1. / ratio) > LLZ_RATIO_MAX) {
        llzs_set_error("llz_resample_mc_init: ratio %d/%d outside 1/%d..%d", L, M, LLZ_RATIO_MAX, LLZ_RATIO_MAX);
        return LLZ_BAD_HANDLE;
    }
    rsm_t *r = (rsm_t *)calloc(1, sizeof(*r));
    if (!r) return LLZ_BAD_HANDLE;
    r->tag = LLZ_TAG_RSM;
    r->device = llzs_device_get();
    r->channels = channels; r->L = L; r->M = M; r->fmt = pcm_format; r->gain = gain;
    const double fc = (1. / L < 1. / M) ? 1. / L : 1. / M;
    int rc = tapmat_build(&r->taps, L, M, fc, L, win_type);
    if (rc == LLZ_OK) {
        r->Q = r->taps.cols;
        const size_t hist_bytes = rsm_sample_bytes(r) * (size_t)channels * (size_t)(r->Q > 1 ? r->Q - 1 : 1);
        r->d_mat = llzs_malloc((r->fmt == LLZ_PCM_I16 ? sizeof(double) : sizeof(float)) * (size_t)L * r->Q);
        r->d_hist[0] = llzs_malloc(hist_bytes);
        r->d_hist[1] = llzs_malloc(hist_bytes);
        if (!r->d_mat || !r->d_hist[0] || !r->d_hist[1]) rc = LLZ_ERR_NOMEM;
        if (rc == LLZ_OK) rc = rsm_upload_matrix(r);
        if (rc == LLZ_OK && r->fmt == LLZ_PCM_I16_FAST && !llzs_fir_mfma_i16_fits(r->Q, r->M)) {
            llzs_set_error("llz_resample_mc_init: %d taps at 1:%d do not fit the matrix-core kernel", r->Q, r->M);
            rc = LLZ_ERR_RANGE;
        }
        if (rc == LLZ_OK) rc = llzs_memset(r->d_hist[0], 0, hist_bytes, NULL);
        if (rc == LLZ_OK) rc = llzs_memset(r->d_hist[1], 0, hist_bytes, NULL);
        if (rc == LLZ_OK) rc = llzs_sync(NULL);
    }
    if (rc != LLZ_OK) {
        rsm_destroy(r);
        return LLZ_BAD_HANDLE;
    }
    return (unsigned long)r;
}

void llz_resample_mc_uninit(unsigned long handle)
{
    if (LLZ_HANDLE_OK(handle, rsm_t, LLZ_TAG_RSM)) {
        const int prev = llzs_device_enter(((rsm_t *)handle)->device);
        llzs_sync(((rsm_t *)handle)->stream);
        rsm_destroy((rsm_t *)handle);
        llzs_device_leave(prev);
    }
}

int llz_resample_mc_sub_len(unsigned long handle)
{
    return LLZ_HANDLE_OK(handle, rsm_t, LLZ_TAG_RSM) ? ((rsm_t *)handle)->Q : LLZ_ERR_ARG;
}

long llz_resample_mc_out_len(unsigned long handle, long n_in)
{
    if (!LLZ_HANDLE_OK(handle, rsm_t, LLZ_TAG_RSM) || n_in < 1) return LLZ_ERR_ARG;
    const rsm_t *r = (const rsm_t *)handle;
    if ((n_in * r->L) % r->M) {
        llzs_set_error("llz_resample_mc: n_in*L = %ld*%d is not a multiple of M = %d", n_in, r->L, r->M);
        return LLZ_ERR_ARG;
    }
    return (n_in * r->L) / r->M;
}

int llz_resample_mc_set_stream(unsigned long handle, void *stream)
{
    if (!LLZ_HANDLE_OK(handle, rsm_t, LLZ_TAG_RSM)) return LLZ_ERR_ARG;
    ((rsm_t *)handle)->stream = stream;
    return LLZ_OK;
}

int llz_resample_mc_get_matrix(unsigned long handle, double *dst, int capacity)
{
    if (!LLZ_HANDLE_OK(handle, rsm_t, LLZ_TAG_RSM) || !dst) return LLZ_ERR_ARG;
    const rsm_t *r = (const rsm_t *)handle;
    const int count = r->L * r->Q;
    if (capacity < count) {
        llzs_set_error("llz_resample_mc_get_matrix: capacity %d < %d", capacity, count);
        return LLZ_ERR_ARG;
    }
    memcpy(dst, r->taps.mat, sizeof(double) * (size_t)count);
    return count;
}

int llz_resample_mc_set_matrix(unsigned long handle, const double *src, int count)
{
    if (!LLZ_HANDLE_OK(handle, rsm_t, LLZ_TAG_RSM) || !src) return LLZ_ERR_ARG;
    rsm_t *r = (rsm_t *)handle;
    if (count != r->L * r->Q) {
        llzs_set_error("llz_resample_mc_set_matrix: %d values, expected %d", count, r->L * r->Q);
        return LLZ_ERR_ARG;
    }
    memcpy(r->taps.mat, src, sizeof(double) * (size_t)count);
    const int prev = llzs_device_enter(r->device);
    int rc = llzs_sync(r->stream);
    if (rc == LLZ_OK) rc = rsm_upload_matrix(r);
    llzs_device_leave(prev);
    return rc;
}

static long rsm_process(rsm_t *r, unsigned long handle, const void *in, long n_in, void *out);

long llz_resample_mc(unsigned long handle, const void *in, long n_in, void *out)
{
    if (!LLZ_HANDLE_OK(handle, rsm_t, LLZ_TAG_RSM) || !in || !out) {
        llzs_set_error("llz_resample_mc: bad handle or NULL buffer");
        return LLZ_ERR_ARG;
    }
    rsm_t *r = (rsm_t *)handle;
    const int prev = llzs_device_enter(r->device);
    const long rc = rsm_process(r, handle, in, n_in, out);
    llzs_device_leave(prev);
    return rc;
}

static long rsm_process(rsm_t *r, unsigned long handle, const void *in, long n_in, void *out)
{
    const long n_out = llz_resample_mc_out_len(handle, n_in);
    if (n_out < 1) return LLZ_ERR_ARG;
    /* calls must start on an L/M period boundary so that (i*M)/L stays exact across calls */
    if ((r->in_count * r->L) % r->M) {
        llzs_set_error("llz_resample_mc: stream position is not on an L/M boundary");
        return LLZ_ERR_ARG;
    }
    const size_t sb = rsm_sample_bytes(r);
    const size_t in_bytes = sb * (size_t)r->channels * (size_t)n_in;
    const size_t out_bytes = sb * (size_t)r->channels * (size_t)n_out;
    const int in_dev = llzs_is_device_ptr(in), out_dev = llzs_is_device_ptr(out);
    if (in_dev < 0 || out_dev < 0) return LLZ_ERR_ARG;            /* a buffer of another GPU: refused, message set */
    const void *d_in = in;
    void *d_out = out;
    int rc = LLZ_OK;
    if (!in_dev) {
        d_in = llz_stage_reserve(&r->st_in, in_bytes);
        if (!d_in) return LLZ_ERR_NOMEM;
        rc = llzs_h2d((void *)d_in, in, in_bytes, r->stream);
    }
    if (rc == LLZ_OK && !out_dev) {
        d_out = llz_stage_reserve(&r->st_out, out_bytes);
        if (!d_out) return LLZ_ERR_NOMEM;
    }
    const void *hist = r->Q > 1 ? r->d_hist[r->cur] : NULL;
    if (rc == LLZ_OK) {
        if (r->fmt == LLZ_PCM_I16_FAST && r->use_screen &&
            (rc = llzs_fir_mfma_i16x((const short *)d_in, (short *)d_out, (const short *)hist, r->d_digits, r->d_gd, r->channels,
                                     n_in, n_out, n_in, n_out, r->Q, r->M, r->screen_shift, r->screen_bias, r->gain,
                                     r->screen_eps, r->stream)) != LLZ_ERR_RANGE)
            ;                                   /* (the exact result is within the format's 1 LSB) */
        else if (r->fmt == LLZ_PCM_I16_FAST)
            rc = llzs_fir_mfma_i16((const short *)d_in, (short *)d_out, (const short *)hist, (const float *)r->d_mat,
                                   r->channels, n_in, n_out, n_in, n_out, r->Q, r->M, (float)r->gain, r->stream);
        else if (r->fmt == LLZ_PCM_I16 && r->use_screen &&
                 (rc = llzs_fir_mfma_i16x((const short *)d_in, (short *)d_out, (const short *)hist, r->d_digits,
                                          (const double *)r->d_mat, r->channels, n_in, n_out, n_in, n_out, r->Q, r->M,
                                          r->screen_shift, r->screen_bias, r->gain, r->screen_eps, r->stream)) !=
                     LLZ_ERR_RANGE)
            ;                                   /* (LLZ_ERR_RANGE: a frame too short or misaligned for the screened kernel) */
        else if (r->fmt == LLZ_PCM_I16 && r->use_screen_lm && r->in_count % r->M == 0 && r->out_count % r->L == 0 &&
                 r->channels <= 65535 &&
                 (rc = llzs_resample_i16x((const short *)d_in, (short *)d_out, (const short *)hist, r->d_scr_atab, r->d_scr_aoff,
                                          r->d_scr_bq, (const double *)r->d_mat, r->channels, n_in, n_out, n_in, n_out, r->L,
                                          r->M, r->Q, r->screen_shift, r->gain, r->screen_eps, r->screen_any_exact,
                                          r->stream)) != LLZ_ERR_RANGE)
            ;                                   /* (LLZ_ERR_RANGE: a frame shorter than one span's image) */
        else if (r->fmt == LLZ_PCM_I16)
            rc = llzs_resample_i16((const short *)d_in, (short *)d_out, (const short *)hist,
                                   (const double *)r->d_mat, r->channels, n_in, n_out, n_in, n_out, r->L, r->M,
                                   r->Q, r->gain, r->out_count, r->in_count, r->stream);
        else if (r->use_mfma)
            rc = llzs_fir_mfma_f32((const float *)d_in, (float *)d_out, (const float *)hist, (const float *)r->d_mat,
                                   r->channels, n_in, n_out, n_in, n_out, r->Q, r->M, (float)r->gain, r->stream);
        else if (r->d_phase)
            rc = llzs_resample_dec_f32((const float *)d_in, (float *)d_out, (const float *)hist, r->d_phase,
                                       r->channels, n_in, n_out, n_in, n_out, r->M, r->Q, r->tp, (float)r->gain,
                                       r->stream);
        else if (r->d_band && r->in_count % r->M == 0 && r->out_count % r->L == 0 && r->channels <= 65535)
            rc = llzs_resample_mfma_f32((const float *)d_in, (float *)d_out, (const float *)hist, r->d_band, r->d_band_c0,
                                        r->channels, n_in, n_out, n_in, n_out, r->L, r->M, r->Q, r->stream);
        else
            rc = llzs_resample_f32((const float *)d_in, (float *)d_out, (const float *)hist,
                                   (const float *)r->d_mat, r->channels, n_in, n_out, n_in, n_out, r->L, r->M,
                                   r->Q, (float)r->gain, r->out_count, r->in_count, r->stream);
    }
    if (rc == LLZ_OK && r->Q > 1) {
        if (r->fmt != LLZ_PCM_F32)
            rc = llzs_tail_i16((const short *)d_in, (const short *)r->d_hist[r->cur],
                               (short *)r->d_hist[r->cur ^ 1], r->channels, n_in, n_in, r->Q - 1, r->stream);
        else
            rc = llzs_fir_tail_f32((const float *)d_in, (const float *)r->d_hist[r->cur],
                                   (float *)r->d_hist[r->cur ^ 1], r->channels, (int)n_in, n_in, r->Q, r->stream);
        if (rc == LLZ_OK) r->cur ^= 1;
    }
    if (rc == LLZ_OK && !out_dev) rc = llzs_d2h(out, d_out, out_bytes, r->stream);
    if (rc != LLZ_OK) return rc;
    r->in_count += n_in;
    r->out_count += n_out;
    return n_out;
}
