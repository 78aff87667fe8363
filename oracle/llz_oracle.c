/*
 * oracle/llz_oracle.c -- CPU restatement of the libllzfilter hot path.  TEST INFRASTRUCTURE ONLY
 * (see llz_oracle.h: never linked into, imported by or called from the product).
 *
 * Parity: PINNED bit-for-bit against the compiled reference (oracle/_ref) and tests/golden/.
 * Build with -O2 -ffp-contract=off (the reference's default build has no FMA contraction on x86-64).
 *
 * Expression trees in the design functions are kept in the reference's association order on purpose:
 * the taps feed bit-exact comparisons, and double arithmetic is not associative.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "llz_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ windows (llz_fir.c:61-158) */

int orc_hamming(double *w, int n)
{
    /* llz_fir.c:61-71: filled from both ends, i <= j */
    for (int i = 0, j = n - 1; i <= j; i++, j--)
        w[j] = w[i] = 0.54 - 0.46 * cos(2 * M_PI * i / (n - 1));
    return n;
}

int orc_blackman(double *w, int n)
{
    /* llz_fir.c:73-83 */
    for (int i = 0, j = n - 1; i <= j; i++, j--)
        w[j] = w[i] = 0.42 - 0.5 * cos(2 * M_PI * i / (n - 1)) + 0.08 * cos(4 * M_PI * i / (n - 1));
    return n;
}

/* modified Bessel I0 by its power series, stop when the term drops under 1e-16 * sum (llz_fir.c:85-103) */
static double bessel_i0(double x)
{
    double half = 0.5 * x, sum = 1.0, p = 1.0, term = 1.0;
    int k = 0;
    while (term > sum * 1E-16) {
        ++k;
        p = p * (half / k);
        term = p * p;
        sum = sum + term;
    }
    return sum;
}

int orc_kaiser_beta(double *w, int n, double beta)
{
    /* llz_fir.c:141-158: no symmetry trick, every point evaluated */
    double denom = bessel_i0(beta);
    for (int i = 0; i < n; i++) {
        double x = (2. * i / (n - 1)) - 1;
        w[i] = bessel_i0(beta * sqrt(1. - x * x)) / denom;
    }
    return n;
}

int orc_kaiser(double *w, int n)
{
    return orc_kaiser_beta(w, n, 8.96); /* llz_fir.c:121-139, beta fixed */
}

double orc_kaiser_atten2beta(double atten)
{
    /* llz_fir.c:105-118 */
    if (atten <= 21.)
        return 0.;
    if (atten < 50.)
        return 0.5842 * pow(atten - 21., 0.4) + 0.07886 * (atten - 21.);
    return 0.1102 * (atten - 8.7);
}

int orc_hamming_cof_num(double ftrans)  { return (int)(6.2 / ftrans); }   /* llz_fir.c:173-176 */
int orc_blackman_cof_num(double ftrans) { return (int)(6.6 / ftrans); }   /* llz_fir.c:178-181 */

int orc_kaiser_cof_num(double ftrans, double atten)
{
    /* llz_fir.c:183-193 */
    if (atten <= 21.)
        return (int)((0.9222 * 2.) / ftrans);
    return (int)(((atten - 7.95) * 2.) / (14.36 * ftrans));
}

/* ------------------------------------------------------------------ tap design (llz_fir.c:39-59, 201-393) */

static double sinc_pi(double x)
{
    /* llz_fir.c:39-59: 1 at 0, exactly 0 at the other integers, else sin(pi*fmod(x,2))/(pi*x) */
    if (x == 0.0)
        return 1.0;
    if (x == floor(x))
        return 0.0;
    return sin(M_PI * fmod(x, 2.0)) / (M_PI * x);
}

static void make_window(double *w, int n, int win)
{
    switch (win) {
    case ORC_HAMMING:  orc_hamming(w, n);  break;
    case ORC_BLACKMAN: orc_blackman(w, n); break;
    case ORC_KAISER:   orc_kaiser(w, n);   break;
    default: for (int i = 0; i < n; i++) w[i] = 0.0; break; /* reference leaves w uninitialised */
    }
}

int orc_fir_design(int kind, double *h, int n, double fc1, double fc2, int win)
{
    if (kind != ORC_LPF && !(n & 1))
        n = n + 1;                                   /* llz_fir.c:305-307, 337-339, 369-371 */
    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    make_window(w, n, win);

    if (kind == ORC_LPF) {
        /* llz_fir.c:201-215: real-valued delay, even n allowed */
        double delay = (double)(n - 1) / 2;
        for (int i = 0, j = n - 1; i <= delay; i++, j--)
            h[j] = h[i] = fc1 * sinc_pi(fc1 * (i - delay)) * w[i];
    } else {
        int delay = (n - 1) / 2;
        for (int i = 0, j = n - 1; i <= delay; i++, j--) {
            double v;
            if (kind == ORC_HPF)                     /* llz_fir.c:225-228 */
                v = -fc1 * sinc_pi(fc1 * (i - delay)) * w[i];
            else if (kind == ORC_BPF)                /* llz_fir.c:243-246 */
                v = (fc2 * sinc_pi(fc2 * (i - delay)) - fc1 * sinc_pi(fc1 * (i - delay))) * w[i];
            else                                     /* llz_fir.c:261-264 */
                v = -(fc2 * sinc_pi(fc2 * (i - delay)) - fc1 * sinc_pi(fc1 * (i - delay))) * w[i];
            h[j] = h[i] = v;
        }
        if (kind == ORC_HPF)      h[delay] = 1 - fc1;            /* :229 */
        else if (kind == ORC_BPF) h[delay] = fc2 - fc1;          /* :247 */
        else                      h[delay] = 1 - (fc2 - fc1);    /* :265 */
    }
    free(w);
    return n;
}

/* ------------------------------------------------------------------ streaming FIR (llz_fir.c:411-625) */

double orc_conv(const double *x, const double *h, int h_len)
{
    /* llz_fir.c:411-426: y = sum_i h[i]*x[n-i], i ascending, plain multiply then add */
    double y = 0.0;
    for (int i = 0; i < h_len; i++)
        y += h[i] * x[-i];
    return y;
}

typedef struct {
    int flt_len, frame_len, buf_len, alloc_len;
    double *h, *buf;
} orc_fir_t;

static void *fir_finish(orc_fir_t *f, int frame_len)
{
    f->frame_len = frame_len;
    f->buf_len = f->flt_len + frame_len - 1;                     /* llz_fir.c:457 */
    /* the reference's flush reads up to buf[2*flt_len-3] (llz_fir.c:611-622) which overruns buf_len when
     * flt_len-2 >= frame_len; the restatement owns enough zeroed storage for that read to be defined */
    f->alloc_len = f->buf_len;
    if (f->alloc_len < 2 * f->flt_len) f->alloc_len = 2 * f->flt_len;
    f->buf = (double *)calloc((size_t)f->alloc_len, sizeof(double));
    return f;
}

void *orc_fir_new(int kind, int frame_len, int flt_len, double fc1, double fc2, int win)
{
    orc_fir_t *f = (orc_fir_t *)calloc(1, sizeof(*f));
    f->h = (double *)malloc(sizeof(double) * (size_t)(flt_len + 1));
    f->flt_len = orc_fir_design(kind, f->h, flt_len, fc1, fc2, win);   /* stored length = returned length */
    return fir_finish(f, frame_len);
}

void *orc_fir_new_taps(int frame_len, const double *h, int flt_len)
{
    orc_fir_t *f = (orc_fir_t *)calloc(1, sizeof(*f));
    f->h = (double *)malloc(sizeof(double) * (size_t)flt_len);
    memcpy(f->h, h, sizeof(double) * (size_t)flt_len);
    f->flt_len = flt_len;
    return fir_finish(f, frame_len);
}

int orc_fir_flt_len(void *p) { return ((orc_fir_t *)p)->flt_len; }
const double *orc_fir_taps(void *p) { return ((orc_fir_t *)p)->h; }

int orc_fir_run(void *p, const double *in, double *out, int frame_len)
{
    orc_fir_t *f = (orc_fir_t *)p;
    if (frame_len > f->frame_len)
        return -1;                                   /* reference: assert (llz_fir.c:559) */
    int hist = f->flt_len - 1;
    /* llz_fir.c:562-566: the tail is taken from the INIT frame length's offset, so the
     * state machine is only a correct streaming filter when every call uses the init frame length */
    int from = f->buf_len - f->flt_len + 1;
    for (int i = 0; i < hist; i++)
        f->buf[i] = f->buf[from + i];
    for (int i = 0; i < frame_len; i++)
        f->buf[hist + i] = in[i];
    for (int i = 0; i < frame_len; i++)
        out[i] = orc_conv(f->buf + hist + i, f->h, f->flt_len);
    return frame_len;
}

int orc_fir_flush(void *p, double *out)
{
    orc_fir_t *f = (orc_fir_t *)p;
    int hist = f->flt_len - 1;
    int from = f->buf_len - f->flt_len + 1;          /* llz_fir.c:605-609 */
    for (int i = 0; i < hist; i++)
        f->buf[i] = f->buf[from + i];
    for (int i = 0; i < f->frame_len; i++)
        f->buf[hist + i] = 0;
    for (int i = 0; i < hist; i++)                   /* llz_fir.c:612-622 */
        out[i] = orc_conv(f->buf + hist + i, f->h, f->flt_len);
    return hist;
}

void orc_fir_free(void *p)
{
    orc_fir_t *f = (orc_fir_t *)p;
    if (!f) return;
    free(f->h); free(f->buf); free(f);
}

/* ------------------------------------------------------------------ IIR (llz_iir.c:37-156) */

typedef struct {
    int M, N;
    double *a, *b, *x, *y;
} orc_iir_t;

void *orc_iir_new(int M, const double *a, int N, const double *b)
{
    orc_iir_t *f = (orc_iir_t *)calloc(1, sizeof(*f));
    f->M = M; f->N = N;
    f->a = (double *)calloc((size_t)M + 1, sizeof(double));
    f->b = (double *)calloc((size_t)N + 1, sizeof(double));
    f->x = (double *)calloc((size_t)N + 1, sizeof(double));
    f->y = (double *)calloc((size_t)M + 1, sizeof(double));
    for (int i = 0; i <= M; i++) f->a[i] = a[i];
    if (b) for (int i = 0; i <= N; i++) f->b[i] = b[i];          /* b == NULL -> zeros (llz_iir.c:54-59) */
    return f;
}

static double iir_step(orc_iir_t *f, double x_in)
{
    /* llz_iir.c:103-132: shift both delay lines, feed-forward sum first, then subtract the feedback
     * terms one by one; a[0] never used */
    int N = f->N, M = f->M;
    for (int i = 0; i < N; i++) f->x[i] = f->x[i + 1];
    f->x[N] = x_in;
    for (int i = 0; i < M; i++) f->y[i] = f->y[i + 1];
    double acc = 0.;
    for (int k = 0; k <= N; k++) acc += f->b[k] * f->x[N - k];
    for (int k = 1; k <= M; k++) acc -= f->a[k] * f->y[M - k];
    f->y[M] = acc;
    return acc;
}

int orc_iir_run(void *p, const double *x, double *y, int frame_len)
{
    for (int i = 0; i < frame_len; i++)
        y[i] = iir_step((orc_iir_t *)p, x[i]);
    return frame_len;
}

int orc_iir_flush(void *p, double *y)
{
    orc_iir_t *f = (orc_iir_t *)p;
    for (int i = 0; i < f->N; i++)                   /* llz_iir.c:147-156 */
        y[i] = iir_step(f, 0.0);
    return f->N;
}

void orc_iir_free(void *p)
{
    orc_iir_t *f = (orc_iir_t *)p;
    if (!f) return;
    free(f->a); free(f->b); free(f->x); free(f->y); free(f);
}

/* ------------------------------------------------------------------ resample family (llz_resample.c) */

typedef struct {
    int mode;            /* 0 decimate, 1 interp, 2 rational */
    int L, M;
    int n;               /* prototype taps */
    int rows, cols;      /* coefficient matrix shape */
    double *h;           /* prototype */
    double *mat;         /* rows x cols */
    double gain;         /* output gain applied per sample */
    int num_in, num_out;
    long long out_index;
    int hist;            /* samples of history kept in front of the frame */
    short *buf;          /* hist + num_in */
} orc_rs_t;

static int igcd(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

static int proto_len_estimate(double ftrans, int win)
{
    switch (win) {                                   /* llz_resample.c:135-148 / 204-215 */
    case ORC_HAMMING:  return orc_hamming_cof_num(ftrans);
    case ORC_BLACKMAN: return orc_blackman_cof_num(ftrans);
    case ORC_KAISER:   return orc_kaiser_cof_num(ftrans, 90);
    }
    return 0;
}

/* shared by polyphase_filter_init (llz_resample.c:124-176) and timevary_filter_init (:193-255):
 * phases = m (polyphase) or L (time-varying); step = phase index multiplier (1 for polyphase: h[m*j+i];
 * M for time-varying: h[j*L + (i*M)%L]) */
static void build_matrix(orc_rs_t *r, int phases, int step_M, double fc, double mgain, int win)
{
    if (mgain == 0) mgain = 1.0;
    int n0 = proto_len_estimate(0.15 * fc, win);
    int half = n0 / (2 * phases);
    r->n = 2 * half * phases + 1;
    r->rows = phases;
    r->cols = r->n / phases + 1;
    r->h = (double *)malloc(sizeof(double) * (size_t)(r->n + 1));
    orc_fir_design(ORC_LPF, r->h, r->n, fc, 0.0, win);
    r->mat = (double *)calloc((size_t)r->rows * r->cols, sizeof(double));
    for (int i = 0; i < r->rows; i++)
        for (int j = 0; j < r->cols; j++) {
            int u = j * phases + (step_M ? (i * step_M) % phases : i);
            r->mat[(size_t)i * r->cols + j] = (u < r->n) ? mgain * r->h[u] : 0;
        }
}

void *orc_rs_new(int mode, int L, int M, double gain, int win)
{
    orc_rs_t *r = (orc_rs_t *)calloc(1, sizeof(*r));
    r->mode = mode; r->gain = gain;
    if (mode == 0) {                                 /* llz_decimate_init, llz_resample.c:271-301 */
        if (M > 16) { free(r); return NULL; }
        r->L = 1; r->M = M;
        build_matrix(r, M, 0, 1. / M, 1, win);
        int m = 1024 / M;
        r->num_in = m * M; r->num_out = m;
        r->hist = r->n;
    } else if (mode == 1) {                          /* llz_interp_init, :320-349 */
        if (L > 16) { free(r); return NULL; }
        r->L = L; r->M = 1;
        build_matrix(r, L, 0, 1. / L, L, win);
        r->num_in = 1024; r->num_out = 1024 * L;
        r->hist = 0;
    } else {                                         /* llz_resample_filter_init, :367-407 */
        double ratio = ((double)L) / M;
        if (ratio > 16 || (1. / ratio) > 16) { free(r); return NULL; }
        r->L = L; r->M = M;
        double fc = (1. / L < 1. / M) ? 1. / L : 1. / M;
        build_matrix(r, L, M, fc, L, win);
        r->num_in = (L * M) / igcd(L, M);
        while (r->num_in < 1024) r->num_in *= 2;
        r->num_out = (r->num_in * L) / M;
        r->hist = r->cols;
    }
    /* + cols of zeroed slack behind the frame: llz_interp reads x[i+k] past the frame end
     * (llz_resample.c:515-523); the restatement defines those reads as zeros */
    r->buf = (short *)calloc((size_t)r->hist + r->num_in + r->cols + 1, sizeof(short));
    return r;
}

int orc_rs_bytes_in(void *p)  { return 2 * ((orc_rs_t *)p)->num_in; }
int orc_rs_bytes_out(void *p) { return 2 * ((orc_rs_t *)p)->num_out; }
int orc_rs_num_taps(void *p)  { return ((orc_rs_t *)p)->n; }
int orc_rs_sub_len(void *p)   { return ((orc_rs_t *)p)->cols; }
const double *orc_rs_matrix(void *p) { return ((orc_rs_t *)p)->mat; }
const double *orc_rs_proto(void *p)  { return ((orc_rs_t *)p)->h; }

static short clamp_trunc(double y)
{
    if (y > 32767)  y = 32767;                       /* llz_resample.c:596-601 */
    if (y < -32768) y = -32768;
    return (short)y;                                 /* C conversion: toward zero */
}

int orc_rs_run(void *p, const unsigned char *in, int in_bytes, unsigned char *out, int *out_bytes)
{
    orc_rs_t *r = (orc_rs_t *)p;
    if (in_bytes != 2 * r->num_in)
        return -1;                                   /* reference: assert */
    const short *src = (const short *)in;
    short *dst = (short *)out;
    short *x = r->buf + r->hist;                     /* first sample of this frame */
    for (int i = 0; i < r->hist; i++)                /* keep the last `hist` samples in front */
        r->buf[i] = r->buf[r->num_in + i];
    memcpy(x, src, sizeof(short) * (size_t)r->num_in);

    if (r->mode == 0) {
        /* llz_decimate, llz_resample.c:457-483: y = sum_m sum_k x[iM + m + Mk - n] * p[m][k] */
        int M = r->M, K = r->cols, n = r->n;
        for (int i = 0; i < r->num_out; i++) {
            double y = 0.0;
            const short *xi = x + (size_t)i * M;
            for (int m = 0; m < M; m++)
                for (int k = 0; k < K; k++)
                    y += xi[m + M * k - n] * r->mat[(size_t)m * K + k];
            y *= r->gain;
            dst[i] = clamp_trunc(y);
        }
    } else if (r->mode == 1) {
        /* llz_interp, :515-536: no history; output slot order reversed within each input sample */
        int L = r->L, K = r->cols;
        for (int i = 0; i < r->num_in; i++)
            for (int m = 0; m < L; m++) {
                double y = 0.0;
                for (int k = 0; k < K; k++)
                    y += x[i + k] * r->mat[(size_t)m * K + k];
                y *= r->gain;
                dst[i * L + (L - 1 - m)] = clamp_trunc(y);
            }
    } else {
        /* llz_resample, :583-603 */
        int L = r->L, M = r->M, Q = r->cols;
        for (int i = 0; i < r->num_out; i++) {
            const short *xp = x + (i * M) / L;
            int l = (int)(r->out_index % L);
            r->out_index++;
            const double *g = r->mat + (size_t)l * Q;
            double y = 0.0;
            for (int k = 0; k < Q; k++)
                y += xp[-k] * g[k];
            y *= r->gain;
            dst[i] = clamp_trunc(y);
        }
    }
    *out_bytes = 2 * r->num_out;
    return 0;
}

void orc_rs_free(void *p)
{
    orc_rs_t *r = (orc_rs_t *)p;
    if (!r) return;
    free(r->h); free(r->mat); free(r->buf); free(r);
}

/* ------------------------------------------------------------------ float FFT (llz_fft.c) */

typedef struct {
    int size, base;
    int *rev;
    double *work, *c, *s;
} orc_fft_t;

static void bitrev_table(int *rev, int size)
{
    /* llz_fft.c:33-49: Gray-code style incremental bit reversal */
    int r = 0, s = 0, i = 0;
    do {
        rev[i++] = s;
        r += 2;
        s ^= size - (size / (r & -r));
    } while (r < (size << 1));
}

static int ceil_log2(int size)
{
    int base = (int)(log(size) / log(2));           /* llz_fft.c:211-214 */
    if ((1 << base) < size) base += 1;
    return base;
}

void *orc_fft_new(int size)
{
    orc_fft_t *f = (orc_fft_t *)calloc(1, sizeof(*f));
    f->size = size; f->base = ceil_log2(size);
    f->rev = (int *)malloc(sizeof(int) * (size_t)size);
    f->work = (double *)malloc(sizeof(double) * 2 * (size_t)size);
    f->c = (double *)malloc(sizeof(double) * (size_t)size);
    f->s = (double *)malloc(sizeof(double) * (size_t)size);
    bitrev_table(f->rev, size);
    for (int i = 0; i < size; i++) {
        double ang = (double)(2 * M_PI * i) / size;  /* llz_fft.c:223-227 */
        f->c[i] = cos(ang);
        f->s[i] = sin(ang);
    }
    return f;
}

static void permute_f64(orc_fft_t *f, double *d, int divide)
{
    int n = f->size;
    for (int i = 0; i < n; i++) {
        int b = f->rev[i];
        f->work[2 * i] = d[2 * b]; f->work[2 * i + 1] = d[2 * b + 1];
    }
    for (int i = 0; i < n; i++) {
        d[2 * i]     = divide ? f->work[2 * i] / n     : f->work[2 * i];
        d[2 * i + 1] = divide ? f->work[2 * i + 1] / n : f->work[2 * i + 1];
    }
}

void orc_fft_fwd(void *p, double *d)
{
    orc_fft_t *f = (orc_fft_t *)p;
    long n = f->size;
    /* llz_fft.c:61-89: DIF, span halves each stage, twiddle index step doubles, w = cos - j sin */
    long tstep = 1;
    for (long span = n; span > 1; span >>= 1, tstep += tstep) {
        long halfc = span / 2;                       /* complex elements per half */
        for (long blk = 0; blk < n; blk += span)
            for (long q = 0; q < halfc; q++) {
                /* the reference's dl counts DOUBLES between partners, i.e. span/2 complex elements */
                double *u = d + 2 * (blk + q), *v = u + span;
                double wr = f->c[q * tstep], wi = -f->s[q * tstep];
                double xr = u[0] + v[0], xi = u[1] + v[1];
                double dr = u[0] - v[0], di = u[1] - v[1];
                u[0] = xr; u[1] = xi;
                v[0] = dr * wr - di * wi;
                v[1] = dr * wi + di * wr;
            }
    }
    permute_f64(f, d, 0);                            /* llz_fft.c:155-163 */
}

void orc_fft_inv(void *p, double *d)
{
    orc_fft_t *f = (orc_fft_t *)p;
    long n = f->size;
    permute_f64(f, d, 1);                            /* llz_fft.c:182-195: gather, then divide by size */
    /* llz_fft.c:101-130: DIT, span doubles each stage, w = cos + j sin */
    long tstep = n >> 1;
    for (long span = 2; tstep > 0; span += span, tstep >>= 1) {
        long halfc = span / 2;
        for (long blk = 0; blk < n; blk += span)
            for (long q = 0; q < halfc; q++) {
                double *u = d + 2 * (blk + q), *v = u + span;
                double wr = f->c[q * tstep], wi = f->s[q * tstep];
                double xr = u[0], xi = u[1], yr = v[0], yi = v[1];
                double dr = yr * wr - yi * wi;
                double di = yr * wi + yi * wr;
                u[0] = xr + dr; u[1] = xi + di;
                v[0] = xr - dr; v[1] = xi - di;
            }
    }
}

void orc_fft_free(void *p)
{
    orc_fft_t *f = (orc_fft_t *)p;
    if (!f) return;
    free(f->rev); free(f->work); free(f->c); free(f->s); free(f);
}

/* ------------------------------------------------------------------ fixed FFT (llz_fft_fixed.c/.h) */

typedef struct {
    int size, base;
    int *rev, *work;
    short *c, *s;
} orc_fftx_t;

static short q15(double v)
{
    /* llz_fft_fixed.h:42-66: round half away from zero of v*2^15, saturate to int32, clip to +-32767 */
    double t = v * (double)(1 << 15);
    double r = (t > 0) ? floor(t + 0.5) : ceil(t - 0.5);
    int q = (r > 2147483647.0) ? 2147483647 : (r < -2147483648.0) ? (-2147483647 - 1) : (int)r;
    if (q < -32767) q = -32767;
    if (q > 32767) q = 32767;
    return (short)q;
}

static inline int mul15(int a, int b)
{
    return (int)(((int64_t)a * (int64_t)b) >> 15);   /* llz_fft_fixed.h:67, arithmetic shift = floor */
}

/* the reference adds/subtracts plain ints (overflow is UB there, wraps in practice); wrap explicitly */
static inline int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
static inline int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }

void *orc_fftx_new(int size)
{
    orc_fftx_t *f = (orc_fftx_t *)calloc(1, sizeof(*f));
    f->size = size; f->base = ceil_log2(size);
    f->rev = (int *)malloc(sizeof(int) * (size_t)size);
    f->work = (int *)malloc(sizeof(int) * 2 * (size_t)size);
    f->c = (short *)malloc(sizeof(short) * (size_t)size);
    f->s = (short *)malloc(sizeof(short) * (size_t)size);
    bitrev_table(f->rev, size);
    for (int i = 0; i < size; i++) {
        double ang = (2 * M_PI * i) / size;          /* llz_fft_fixed.c:243-247 */
        f->c[i] = q15(cos(ang));
        f->s[i] = q15(sin(ang));
    }
    return f;
}

const short *orc_fftx_cos(void *p) { return ((orc_fftx_t *)p)->c; }
const short *orc_fftx_sin(void *p) { return ((orc_fftx_t *)p)->s; }

static void permute_i32(orc_fftx_t *f, int *d)
{
    int n = f->size;
    for (int i = 0; i < n; i++) {
        int b = f->rev[i];
        f->work[2 * i] = d[2 * b]; f->work[2 * i + 1] = d[2 * b + 1];
    }
    memcpy(d, f->work, sizeof(int) * 2 * (size_t)n);
}

void orc_fftx_fwd(void *p, int *d)
{
    orc_fftx_t *f = (orc_fftx_t *)p;
    long n = f->size, tstep = 1;
    /* llz_fft_fixed.c:61-95 */
    for (long span = n; span > 1; span >>= 1, tstep += tstep) {
        long halfc = span / 2;
        for (long blk = 0; blk < n; blk += span)
            for (long q = 0; q < halfc; q++) {
                int *u = d + 2 * (blk + q), *v = u + span;
                short wr = f->c[q * tstep], wi = (short)(-f->s[q * tstep]);
                int xr = wadd(u[0], v[0]), xi = wadd(u[1], v[1]);
                int dr = wsub(u[0], v[0]), di = wsub(u[1], v[1]);
                u[0] = xr; u[1] = xi;
                v[0] = wsub(mul15(dr, wr), mul15(di, wi));
                v[1] = wadd(mul15(dr, wi), mul15(di, wr));
            }
    }
    permute_i32(f, d);                               /* llz_fft_fixed.c:165-174 */
}

void orc_fftx_inv(void *p, int *d)
{
    orc_fftx_t *f = (orc_fftx_t *)p;
    long n = f->size;
    permute_i32(f, d);                               /* llz_fft_fixed.c:200-208 */
    long tstep = n >> 1;
    for (long span = 2; tstep > 0; span += span, tstep >>= 1) {   /* :108-140 */
        long halfc = span / 2;
        for (long blk = 0; blk < n; blk += span)
            for (long q = 0; q < halfc; q++) {
                int *u = d + 2 * (blk + q), *v = u + span;
                short wr = f->c[q * tstep], wi = f->s[q * tstep];
                int xr = u[0], xi = u[1], yr = v[0], yi = v[1];
                int dr = wsub(mul15(yr, wr), mul15(yi, wi));
                int di = wadd(mul15(yr, wi), mul15(yi, wr));
                u[0] = wadd(xr, dr); u[1] = wadd(xi, di);
                v[0] = wsub(xr, dr); v[1] = wsub(xi, di);
            }
    }
    for (long i = 0; i < 2 * n; i++)                 /* :212-215: scale only at the very end */
        d[i] = d[i] >> f->base;
}

void orc_fftx_free(void *p)
{
    orc_fftx_t *f = (orc_fftx_t *)p;
    if (!f) return;
    free(f->rev); free(f->work); free(f->c); free(f->s); free(f);
}

/* ------------------------------------------------------------------ correlation (llz_corr.c) */

void orc_autocorr(const double *x, int n, int p, double *r)
{
    for (int k = 0; k <= p; k++) {                   /* llz_corr.c:38-47: plain running sums, ascending i */
        double acc = 0.0;
        for (int i = 0; i + k < n; i++) acc += x[i] * x[i + k];
        r[k] = acc;
    }
}

void orc_crosscorr(const double *x, const double *y, int n, int p, double *r)
{
    for (int k = 0; k <= p; k++) {                   /* llz_corr.c:49-58 */
        double acc = 0.0;
        for (int i = 0; i + k < n; i++) acc += x[i] * y[i + k];
        r[k] = acc;
    }
}

double orc_corr_cof(const double *a, const double *b, int len)
{
    double ab = 0, aa = 0, bb = 0;                   /* llz_corr.c:61-78 */
    for (int k = 0; k < len; k++) {
        ab += a[k] * b[k];
        aa += a[k] * a[k];
        bb += b[k] * b[k];
    }
    return ab / sqrt(aa * bb);
}

typedef struct {
    int fft_len;
    void *fft;
    double *b1, *b2;
} orc_acf_t;

void *orc_acf_new(int n)
{
    /* llz_corr.c:81-119: level = (int)log2(2n), bumped when 2^level < 2n */
    int level = (int)log2((double)(2 * n));
    if ((1 << level) < 2 * n) level += 1;
    orc_acf_t *h = (orc_acf_t *)calloc(1, sizeof(*h));
    h->fft_len = 1 << level;
    h->fft = orc_fft_new(h->fft_len);
    h->b1 = (double *)calloc(2 * (size_t)h->fft_len, sizeof(double));
    h->b2 = (double *)calloc(2 * (size_t)h->fft_len, sizeof(double));
    return h;
}

int orc_acf_fft_len(void *p) { return ((orc_acf_t *)p)->fft_len; }

void orc_acf_run(void *p, const double *x, int n, int pord, double *r)
{
    orc_acf_t *h = (orc_acf_t *)p;
    /* llz_corr.c:155-177: zero-padded forward FFT; power spectrum of the FIRST n BINS ONLY (the rest stays zero --
     * the reference's quirk, kept); inverse FFT; result doubled */
    memset(h->b1, 0, sizeof(double) * 2 * (size_t)h->fft_len);
    for (int i = 0; i < n; i++) h->b1[2 * i] = x[i];
    orc_fft_fwd(h->fft, h->b1);
    memset(h->b2, 0, sizeof(double) * 2 * (size_t)h->fft_len);
    for (int i = 0; i < n; i++)
        h->b2[2 * i] = h->b1[2 * i] * h->b1[2 * i] + h->b1[2 * i + 1] * h->b1[2 * i + 1];
    orc_fft_inv(h->fft, h->b2);
    for (int i = 0; i <= pord; i++) r[i] = h->b2[2 * i] * 2;
}

void orc_acf_free(void *p)
{
    orc_acf_t *h = (orc_acf_t *)p;
    if (!h) return;
    orc_fft_free(h->fft); free(h->b1); free(h->b2); free(h);
}

/* ------------------------------------------------------------------ analysis / synthesis by windowed FFT frames
 * reference libllzfilter/llz_asmodel.c:109-310.  One struct serves both directions, as in the
 * reference (llz_analysis_fft_init and llz_synthesis_fft_init build the same state). */
typedef struct {
    int frame_len, fft_len;
    double *x_buf, *fft_buf, *window;
    void *fft;
    double magic;
} orc_stft_t;

void *orc_stft_new(int overlap_hint, int frame_len, int win)
{
    orc_stft_t *f = (orc_stft_t *)calloc(1, sizeof(*f));
    f->frame_len = frame_len;
    /* llz_asmodel.c:114-123: 3/4 overlap -> 4 frames per transform and the empirical 0.812; 1/2 overlap -> 2 and 1 */
    if (overlap_hint == 0) { f->fft_len = frame_len << 2; f->magic = 0.812; }
    else                   { f->fft_len = frame_len << 1; f->magic = 1; }
    f->x_buf = (double *)calloc((size_t)f->fft_len, sizeof(double));
    f->fft_buf = (double *)calloc(2 * (size_t)f->fft_len, sizeof(double));
    f->window = (double *)calloc((size_t)f->fft_len, sizeof(double));
    f->fft = orc_fft_new(f->fft_len);
    if (win == ORC_HAMMING) orc_hamming(f->window, f->fft_len);            /* llz_asmodel.c:133-143 */
    else if (win == ORC_BLACKMAN) orc_blackman(f->window, f->fft_len);
    else orc_kaiser(f->window, f->fft_len);
    return f;
}

int orc_stft_fft_len(void *p) { return ((orc_stft_t *)p)->fft_len; }

void orc_stft_analysis(void *p, const double *x, double *re, double *im)
{
    orc_stft_t *f = (orc_stft_t *)p;
    const int F = f->frame_len, N = f->fft_len;
    /* llz_asmodel.c:188-204: slide the frame in at the end, window, transform, keep bins 0..N/2 */
    for (int i = 0; i < N - F; i++) f->x_buf[i] = f->x_buf[i + F];
    for (int i = 0; i < F; i++) f->x_buf[i + N - F] = x[i];
    for (int i = 0; i < N; i++) {
        f->fft_buf[2 * i] = f->x_buf[i] * f->window[i];
        f->fft_buf[2 * i + 1] = 0;
    }
    orc_fft_fwd(f->fft, f->fft_buf);
    for (int i = 0; i < (N >> 1) + 1; i++) {
        re[i] = f->fft_buf[2 * i];
        im[i] = f->fft_buf[2 * i + 1];
    }
}

void orc_stft_synthesis(void *p, const double *re, const double *im, double *x)
{
    orc_stft_t *f = (orc_stft_t *)p;
    const int F = f->frame_len, N = f->fft_len;
    /* llz_asmodel.c:279-288: bins 0..N/2 as given, the upper half by Hermitian symmetry */
    for (int i = 0; i < (N >> 1) + 1; i++) {
        f->fft_buf[2 * i] = re[i];
        f->fft_buf[2 * i + 1] = im[i];
    }
    for (int i = 0, j = (N >> 1) - 1; i < (N >> 1) - 1; i++, j--) {
        f->fft_buf[N + 2 + 2 * i] = re[j];
        f->fft_buf[N + 2 + 2 * i + 1] = -im[j];
    }
    orc_fft_inv(f->fft, f->fft_buf);
    /* llz_asmodel.c:292-304: windowed overlap-add, scaled output of the oldest frame_len samples, slide */
    for (int i = 0; i < N; i++) {
        const double t = f->fft_buf[2 * i] * f->window[i];
        f->x_buf[i] = f->x_buf[i] + t;
    }
    for (int i = 0; i < F; i++) x[i] = f->magic * f->x_buf[i];
    for (int i = 0; i < N - F; i++) f->x_buf[i] = f->x_buf[i + F];
    for (int i = 0; i < F; i++) f->x_buf[i + N - F] = 0;
}

void orc_stft_free(void *p)
{
    orc_stft_t *f = (orc_stft_t *)p;
    if (!f) return;
    orc_fft_free(f->fft); free(f->x_buf); free(f->fft_buf); free(f->window); free(f);
}

/* ------------------------------------------------------------------ MDCT
 * reference libllzfilter/llz_mdct.c:97-620: three algorithms behind one handle.
 *   type 0 (MDCT_ORIGIN) the defining sums, type 1 (MDCT_FFT) one N-point FFT, type 2 (MDCT_FFT4) one N/4-point FFT. */
int orc_mdct_sine(double *w, int N)
{
    for (int n = 0; n < N; n++) {                       /* llz_mdct.c:97-108 */
        const double tmp = (M_PI / N) * (n + 0.5);
        w[n] = sin(tmp);
    }
    return N;
}

static double mdct_bessel(double x)                    /* llz_mdct.c:110-129 (same series as llz_fir.c's) */
{
    double xh = (double)0.5 * x, sum = 1.0, pw = 1.0, ds = 1.0;
    int k = 0;
    while (ds > sum * 1E-16) {
        ++k;
        pw = pw * (xh / k);
        ds = pw * pw;
        sum = sum + ds;
    }
    return sum;
}

int orc_mdct_kbd(double *w, int N, double alpha)
{
    /* llz_mdct.c:131-182: Kaiser window of N/2+1 points with beta = alpha*pi, cumulative sums, square root */
    const int N2 = N >> 1;
    double *w1 = (double *)malloc(sizeof(double) * (size_t)(N2 + 1));
    const double beta = alpha * M_PI;
    for (int i = 0; i < N2 + 1; i++) {
        const double Ib = mdct_bessel(beta);
        const double x = (double)((2. * i / (N2 + 1 - 1)) - 1);
        const double Ia = mdct_bessel(beta * (double)sqrt(1. - x * x));
        w1[i] = (double)(Ia / Ib);
    }
    double sum = 0.0, tmp = 0.0;
    for (int i = 0; i < N2 + 1; i++) sum += w1[i];
    sum = 1.0 / sum;
    for (int i = 0, j = N - 1; i < N2; i++, j--) {
        tmp += w1[i];
        w[i] = w[j] = sqrt(tmp * sum);
    }
    free(w1);
    return N;
}

typedef struct {
    int type, length;
    void *fft;
    double *fft_buf;
    double *cos_pos, *cos_inv;                          /* type 0: [N/2][N] and [N][N/2] */
    double *pre_c_pos, *pre_s_pos, *c_pos, *s_pos, *pre_c_inv, *pre_s_inv, *c_inv, *s_inv;   /* type 1 */
    double *tw_c, *tw_s, *rot, sqrt_cof;                /* type 2 */
} orc_mdct_t;

void *orc_mdct_new(int type, int size)
{
    orc_mdct_t *f = (orc_mdct_t *)calloc(1, sizeof(*f));
    int base = (int)(log(size) / log(2));               /* llz_mdct.c:375-379 */
    if ((1 << base) < size) base += 1;
    const int length = 1 << base;
    f->length = length;
    f->type = type;
    if (type == 0) {                                    /* llz_mdct.c:384-403 */
        f->cos_pos = (double *)calloc((size_t)(length >> 1) * length, sizeof(double));
        f->cos_inv = (double *)calloc((size_t)length * (length >> 1), sizeof(double));
        for (int k = 0; k < (length >> 1); k++)
            for (int n = 0; n < length; n++) {
                const double tmp = (M_PI / (2 * length)) * (2 * n + 1 + (length >> 1)) * (2 * k + 1);
                f->cos_pos[(size_t)k * length + n] = f->cos_inv[(size_t)n * (length >> 1) + k] = cos(tmp);
            }
    } else if (type == 1) {                             /* llz_mdct.c:404-448 */
        const double n0 = ((double)length / 2 + 1) / 2;
        f->fft = orc_fft_new(length);
        f->fft_buf = (double *)calloc(2 * (size_t)length, sizeof(double));
        f->pre_c_pos = (double *)malloc(sizeof(double) * length); f->pre_s_pos = (double *)malloc(sizeof(double) * length);
        f->c_pos = (double *)malloc(sizeof(double) * (length >> 1)); f->s_pos = (double *)malloc(sizeof(double) * (length >> 1));
        f->pre_c_inv = (double *)malloc(sizeof(double) * length); f->pre_s_inv = (double *)malloc(sizeof(double) * length);
        f->c_inv = (double *)malloc(sizeof(double) * length); f->s_inv = (double *)malloc(sizeof(double) * length);
        for (int k = 0; k < length; k++) {
            f->pre_c_pos[k] = cos(-(M_PI * k) / length);
            f->pre_s_pos[k] = sin(-(M_PI * k) / length);
        }
        for (int k = 0; k < (length >> 1); k++) {
            f->c_pos[k] = cos(-2 * M_PI * n0 * (k + 0.5) / length);
            f->s_pos[k] = sin(-2 * M_PI * n0 * (k + 0.5) / length);
        }
        for (int k = 0; k < length; k++) {
            f->pre_c_inv[k] = cos((2 * M_PI * k * n0) / length);
            f->pre_s_inv[k] = sin((2 * M_PI * k * n0) / length);
        }
        for (int k = 0; k < length; k++) {
            f->c_inv[k] = cos(M_PI * (k + n0) / length);
            f->s_inv[k] = sin(M_PI * (k + n0) / length);
        }
    } else {                                            /* llz_mdct.c:449-467 */
        f->fft = orc_fft_new(length >> 2);
        f->fft_buf = (double *)calloc((size_t)(length >> 1), sizeof(double));
        f->sqrt_cof = 1. / sqrt(length);
        f->rot = (double *)calloc((size_t)length, sizeof(double));
        f->tw_c = (double *)malloc(sizeof(double) * (length >> 2));
        f->tw_s = (double *)malloc(sizeof(double) * (length >> 2));
        for (int k = 0; k < (length >> 2); k++) {
            f->tw_c[k] = cos(-2 * M_PI * (k + 0.125) / length);
            f->tw_s[k] = sin(-2 * M_PI * (k + 0.125) / length);
        }
    }
    return f;
}

int orc_mdct_length(void *p) { return ((orc_mdct_t *)p)->length; }

void orc_mdct_fwd(void *p, const double *x, double *X)
{
    orc_mdct_t *f = (orc_mdct_t *)p;
    const int N = f->length, N2 = N >> 1, N4 = N >> 2;
    if (f->type == 0) {                                 /* llz_mdct.c:185-202 */
        for (int k = 0; k < N2; k++) {
            double Xk = 0;
            for (int n = 0; n < N; n++) Xk += x[n] * f->cos_pos[(size_t)k * N + n];
            X[k] = Xk;
        }
    } else if (f->type == 1) {                          /* llz_mdct.c:225-241 */
        for (int k = 0; k < N; k++) {
            f->fft_buf[k + k] = x[k] * f->pre_c_pos[k];
            f->fft_buf[k + k + 1] = x[k] * f->pre_s_pos[k];
        }
        orc_fft_fwd(f->fft, f->fft_buf);
        for (int k = 0; k < N2; k++)
            X[k] = f->fft_buf[k + k] * f->c_pos[k] - f->fft_buf[k + k + 1] * f->s_pos[k];
    } else {                                            /* llz_mdct.c:266-303 */
        double *rot = f->rot;
        memset(rot, 0, sizeof(double) * (size_t)f->length);
        for (int k = 0; k < N4; k++) rot[k] = -x[k + 3 * N4];
        for (int k = N4; k < N; k++) rot[k] = x[k - N4];
        for (int k = 0; k < N4; k++) {
            const double re = rot[2 * k] - rot[N - 1 - 2 * k];
            const double im = rot[N2 - 1 - 2 * k] - rot[N2 + 2 * k];
            f->fft_buf[k + k] = 0.5 * (re * f->tw_c[k] - im * f->tw_s[k]);
            f->fft_buf[k + k + 1] = 0.5 * (re * f->tw_s[k] + im * f->tw_c[k]);
        }
        orc_fft_fwd(f->fft, f->fft_buf);
        for (int k = 0; k < N4; k++) {
            const double re = f->fft_buf[k + k], im = f->fft_buf[k + k + 1];
            X[2 * k] = 2 * (re * f->tw_c[k] - im * f->tw_s[k]);
            X[N2 - 1 - 2 * k] = -2 * (re * f->tw_s[k] + im * f->tw_c[k]);
        }
    }
}

void orc_mdct_inv(void *p, const double *X, double *x)
{
    orc_mdct_t *f = (orc_mdct_t *)p;
    const int N = f->length, N2 = N >> 1, N4 = N >> 2;
    if (f->type == 0) {                                 /* llz_mdct.c:204-222 */
        for (int n = 0; n < N; n++) {
            double xn = 0;
            for (int k = 0; k < N2; k++) xn += X[k] * f->cos_inv[(size_t)n * N2 + k];
            x[n] = (xn * 4) / N;
        }
    } else if (f->type == 1) {                          /* llz_mdct.c:243-264 */
        for (int k = 0; k < N2; k++) {
            f->fft_buf[k + k] = X[k] * f->pre_c_inv[k];
            f->fft_buf[k + k + 1] = X[k] * f->pre_s_inv[k];
        }
        for (int k = N2, i = N2 - 1; k < N; k++, i--) {
            f->fft_buf[k + k] = -X[i] * f->pre_c_inv[k];
            f->fft_buf[k + k + 1] = -X[i] * f->pre_s_inv[k];
        }
        orc_fft_inv(f->fft, f->fft_buf);
        for (int k = 0; k < N; k++)
            x[k] = 2 * (f->fft_buf[k + k] * f->c_inv[k] - f->fft_buf[k + k + 1] * f->s_inv[k]);
    } else {                                            /* llz_mdct.c:305-353; NOTE the forward llz_fft here too */
        double *rot = f->rot;
        const double cof = f->sqrt_cof;
        memset(rot, 0, sizeof(double) * (size_t)f->length);
        for (int k = 0; k < N4; k++) {
            const double re = X[2 * k], im = X[N2 - 1 - 2 * k];
            f->fft_buf[k + k] = 0.5 * (re * f->tw_c[k] - im * f->tw_s[k]);
            f->fft_buf[k + k + 1] = 0.5 * (re * f->tw_s[k] + im * f->tw_c[k]);
        }
        orc_fft_fwd(f->fft, f->fft_buf);
        for (int k = 0; k < N4; k++) {
            const double re = f->fft_buf[k + k], im = f->fft_buf[k + k + 1];
            f->fft_buf[k + k] = 8 * cof * (re * f->tw_c[k] - im * f->tw_s[k]);
            f->fft_buf[k + k + 1] = 8 * cof * (re * f->tw_s[k] + im * f->tw_c[k]);
        }
        for (int k = 0; k < N4; k++) {
            rot[2 * k] = f->fft_buf[k + k];
            rot[N2 + 2 * k] = f->fft_buf[k + k + 1];
        }
        for (int k = 1; k < N; k += 2) rot[k] = -rot[N - 1 - k];
        for (int k = 0; k < 3 * N4; k++) x[k] = rot[N4 + k] * cof;
        for (int k = 3 * N4; k < N; k++) x[k] = -rot[k - 3 * N4] * cof;
    }
}

void orc_mdct_free(void *p)
{
    orc_mdct_t *f = (orc_mdct_t *)p;
    if (!f) return;
    if (f->fft) orc_fft_free(f->fft);
    free(f->fft_buf); free(f->cos_pos); free(f->cos_inv);
    free(f->pre_c_pos); free(f->pre_s_pos); free(f->c_pos); free(f->s_pos);
    free(f->pre_c_inv); free(f->pre_s_inv); free(f->c_inv); free(f->s_inv);
    free(f->tw_c); free(f->tw_s); free(f->rot); free(f);
}

/* fixed-point MDCT, reference libllzfilter/llz_mdct_fixed.c:116-392: the float algorithms with Q15 tables
 * (q15() above = LLZ_FIX15) and (int64 a * b) >> 15 products (mul15() above = LLZ_FIXMUL_32X15); int32 adds wrap */
typedef struct {
    int type, length;
    void *fft;
    int *fft_buf, *rot;
    short *cos_pos, *cos_inv;
    short *pre_c_pos, *pre_s_pos, *c_pos, *s_pos, *pre_c_inv, *pre_s_inv, *c_inv, *s_inv;
    short *tw_c, *tw_s, sqrt_cof;
} orc_mdctx_t;

static inline int wneg(int a) { return (int)(0u - (unsigned)a); }
static inline int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

void *orc_mdctx_new(int type, int size)
{
    orc_mdctx_t *f = (orc_mdctx_t *)calloc(1, sizeof(*f));
    int base = (int)(log(size) / log(2));
    if ((1 << base) < size) base += 1;
    const int length = 1 << base;
    f->length = length; f->type = type;
    if (type == 0) {                                    /* llz_mdct_fixed.c:307-325 */
        f->cos_pos = (short *)calloc((size_t)(length >> 1) * length, sizeof(short));
        f->cos_inv = (short *)calloc((size_t)length * (length >> 1), sizeof(short));
        for (int k = 0; k < (length >> 1); k++)
            for (int n = 0; n < length; n++) {
                const double tmp = (M_PI / (2 * length)) * (2 * n + 1 + (length >> 1)) * (2 * k + 1);
                f->cos_pos[(size_t)k * length + n] = f->cos_inv[(size_t)n * (length >> 1) + k] = q15(cos(tmp));
            }
    } else if (type == 1) {                             /* llz_mdct_fixed.c:326-367 */
        const double n0 = ((double)length / 2 + 1) / 2;
        f->fft = orc_fftx_new(length);
        f->fft_buf = (int *)calloc(2 * (size_t)length, sizeof(int));
        f->pre_c_pos = (short *)malloc(sizeof(short) * length); f->pre_s_pos = (short *)malloc(sizeof(short) * length);
        f->c_pos = (short *)malloc(sizeof(short) * (length >> 1)); f->s_pos = (short *)malloc(sizeof(short) * (length >> 1));
        f->pre_c_inv = (short *)malloc(sizeof(short) * length); f->pre_s_inv = (short *)malloc(sizeof(short) * length);
        f->c_inv = (short *)malloc(sizeof(short) * length); f->s_inv = (short *)malloc(sizeof(short) * length);
        for (int k = 0; k < length; k++) {
            f->pre_c_pos[k] = q15(cos(-(M_PI * k) / length));
            f->pre_s_pos[k] = q15(sin(-(M_PI * k) / length));
        }
        for (int k = 0; k < (length >> 1); k++) {
            f->c_pos[k] = q15(cos(-2 * M_PI * n0 * (k + 0.5) / length));
            f->s_pos[k] = q15(sin(-2 * M_PI * n0 * (k + 0.5) / length));
        }
        for (int k = 0; k < length; k++) {
            f->pre_c_inv[k] = q15(cos((2 * M_PI * k * n0) / length));
            f->pre_s_inv[k] = q15(sin((2 * M_PI * k * n0) / length));
        }
        for (int k = 0; k < length; k++) {
            f->c_inv[k] = q15(cos(M_PI * (k + n0) / length));
            f->s_inv[k] = q15(sin(M_PI * (k + n0) / length));
        }
    } else {                                            /* llz_mdct_fixed.c:368-387 */
        f->fft = orc_fftx_new(length >> 2);
        f->fft_buf = (int *)calloc((size_t)(length >> 1), sizeof(int));
        f->sqrt_cof = q15(1. / sqrt(length));
        f->rot = (int *)calloc((size_t)length, sizeof(int));
        f->tw_c = (short *)malloc(sizeof(short) * (length >> 2));
        f->tw_s = (short *)malloc(sizeof(short) * (length >> 2));
        for (int k = 0; k < (length >> 2); k++) {
            f->tw_c[k] = q15(cos(-2 * M_PI * (k + 0.125) / length));
            f->tw_s[k] = q15(sin(-2 * M_PI * (k + 0.125) / length));
        }
    }
    return f;
}

int orc_mdctx_length(void *p) { return ((orc_mdctx_t *)p)->length; }

void orc_mdctx_fwd(void *p, const int *x, int *X)
{
    orc_mdctx_t *f = (orc_mdctx_t *)p;
    const int N = f->length, N2 = N >> 1, N4 = N >> 2;
    if (f->type == 0) {                                 /* llz_mdct_fixed.c:116-133 */
        for (int k = 0; k < N2; k++) {
            int Xk = 0;
            for (int n = 0; n < N; n++) Xk = wadd(Xk, mul15(x[n], f->cos_pos[(size_t)k * N + n]));
            X[k] = Xk;
        }
    } else if (f->type == 1) {                          /* llz_mdct_fixed.c:155-172 */
        for (int k = 0; k < N; k++) {
            f->fft_buf[k + k] = mul15(x[k], f->pre_c_pos[k]);
            f->fft_buf[k + k + 1] = mul15(x[k], f->pre_s_pos[k]);
        }
        orc_fftx_fwd(f->fft, f->fft_buf);
        for (int k = 0; k < N2; k++)
            X[k] = wsub(mul15(f->fft_buf[k + k], f->c_pos[k]), mul15(f->fft_buf[k + k + 1], f->s_pos[k]));
    } else {                                            /* llz_mdct_fixed.c:197-233 */
        int *rot = f->rot;
        memset(rot, 0, sizeof(int) * (size_t)f->length);
        for (int k = 0; k < N4; k++) rot[k] = wneg(x[k + 3 * N4]);
        for (int k = N4; k < N; k++) rot[k] = x[k - N4];
        for (int k = 0; k < N4; k++) {
            const int re = wsub(rot[2 * k], rot[N - 1 - 2 * k]);
            const int im = wsub(rot[N2 - 1 - 2 * k], rot[N2 + 2 * k]);
            f->fft_buf[k + k] = wsub(mul15(re, f->tw_c[k]), mul15(im, f->tw_s[k])) >> 1;
            f->fft_buf[k + k + 1] = wadd(mul15(re, f->tw_s[k]), mul15(im, f->tw_c[k])) >> 1;
        }
        orc_fftx_fwd(f->fft, f->fft_buf);
        for (int k = 0; k < N4; k++) {
            const int re = f->fft_buf[k + k], im = f->fft_buf[k + k + 1];
            X[2 * k] = wmul(2, wsub(mul15(re, f->tw_c[k]), mul15(im, f->tw_s[k])));
            X[N2 - 1 - 2 * k] = wmul(-2, wadd(mul15(re, f->tw_s[k]), mul15(im, f->tw_c[k])));
        }
    }
}

void orc_mdctx_inv(void *p, const int *X, int *x)
{
    orc_mdctx_t *f = (orc_mdctx_t *)p;
    const int N = f->length, N2 = N >> 1, N4 = N >> 2;
    if (f->type == 0) {                                 /* llz_mdct_fixed.c:135-152 */
        for (int n = 0; n < N; n++) {
            int xn = 0;
            for (int k = 0; k < N2; k++) xn = wadd(xn, mul15(X[k], f->cos_inv[(size_t)n * N2 + k]));
            x[n] = wmul(xn, 4) / N;
        }
    } else if (f->type == 1) {                          /* llz_mdct_fixed.c:174-195 */
        for (int k = 0; k < N2; k++) {
            f->fft_buf[k + k] = mul15(X[k], f->pre_c_inv[k]);
            f->fft_buf[k + k + 1] = mul15(X[k], f->pre_s_inv[k]);
        }
        for (int k = N2, i = N2 - 1; k < N; k++, i--) {
            f->fft_buf[k + k] = mul15(wneg(X[i]), f->pre_c_inv[k]);
            f->fft_buf[k + k + 1] = mul15(wneg(X[i]), f->pre_s_inv[k]);
        }
        orc_fftx_inv(f->fft, f->fft_buf);
        for (int k = 0; k < N; k++)
            x[k] = (int)((unsigned)wsub(mul15(f->fft_buf[k + k], f->c_inv[k]), mul15(f->fft_buf[k + k + 1], f->s_inv[k])) << 1);
    } else {                                            /* llz_mdct_fixed.c:235-283 */
        int *rot = f->rot;
        const short cof = f->sqrt_cof;
        memset(rot, 0, sizeof(int) * (size_t)f->length);
        for (int k = 0; k < N4; k++) {
            const int re = X[2 * k], im = X[N2 - 1 - 2 * k];
            f->fft_buf[k + k] = wsub(mul15(re, f->tw_c[k]), mul15(im, f->tw_s[k])) >> 1;
            f->fft_buf[k + k + 1] = wadd(mul15(re, f->tw_s[k]), mul15(im, f->tw_c[k])) >> 1;
        }
        orc_fftx_fwd(f->fft, f->fft_buf);
        for (int k = 0; k < N4; k++) {
            const int re = f->fft_buf[k + k], im = f->fft_buf[k + k + 1];
            int tmp = wsub(mul15(re, f->tw_c[k]), mul15(im, f->tw_s[k]));
            f->fft_buf[k + k] = wmul(8, mul15(tmp, cof));
            tmp = wadd(mul15(re, f->tw_s[k]), mul15(im, f->tw_c[k]));
            f->fft_buf[k + k + 1] = wmul(8, mul15(tmp, cof));
        }
        for (int k = 0; k < N4; k++) {
            rot[2 * k] = f->fft_buf[k + k];
            rot[N2 + 2 * k] = f->fft_buf[k + k + 1];
        }
        for (int k = 1; k < N; k += 2) rot[k] = wneg(rot[N - 1 - k]);
        for (int k = 0; k < 3 * N4; k++) x[k] = mul15(rot[N4 + k], cof);
        for (int k = 3 * N4; k < N; k++) x[k] = mul15(wneg(rot[k - 3 * N4]), cof);
    }
}

void orc_mdctx_free(void *p)
{
    orc_mdctx_t *f = (orc_mdctx_t *)p;
    if (!f) return;
    if (f->fft) orc_fftx_free(f->fft);
    free(f->fft_buf); free(f->rot); free(f->cos_pos); free(f->cos_inv);
    free(f->pre_c_pos); free(f->pre_s_pos); free(f->c_pos); free(f->s_pos);
    free(f->pre_c_inv); free(f->pre_s_inv); free(f->c_inv); free(f->s_inv);
    free(f->tw_c); free(f->tw_s); free(f);
}

/* analysis / synthesis by windowed MDCT frames with 50 % overlap (TDAC), reference llz_asmodel.c:313-463; the
 * transform is always type 2 (MDCT_FFT4) of length 2*frame_len; win 0 = sine, 1 = KBD with alpha 6 */
typedef struct {
    int frame_len, mdct_len;
    double *x_buf, *mdct_buf, *window;
    void *mdct;
} orc_amdct_t;

void *orc_amdct_new(int frame_len, int win)
{
    orc_amdct_t *f = (orc_amdct_t *)calloc(1, sizeof(*f));
    f->frame_len = frame_len;
    f->mdct_len = frame_len << 1;
    f->x_buf = (double *)calloc((size_t)f->mdct_len, sizeof(double));
    f->mdct_buf = (double *)calloc((size_t)f->mdct_len, sizeof(double));
    f->window = (double *)calloc((size_t)f->mdct_len, sizeof(double));
    f->mdct = orc_mdct_new(2, f->mdct_len);
    if (win == 0) orc_mdct_sine(f->window, f->mdct_len);
    else orc_mdct_kbd(f->window, f->mdct_len, 6);
    return f;
}

void orc_amdct_analysis(void *p, const double *x, double *X)
{
    orc_amdct_t *f = (orc_amdct_t *)p;
    const int F = f->frame_len;
    for (int i = 0; i < F; i++) f->x_buf[i] = f->x_buf[i + F];          /* llz_asmodel.c:365-376 */
    for (int i = 0; i < F; i++) f->x_buf[i + F] = x[i];
    for (int i = 0; i < f->mdct_len; i++) f->mdct_buf[i] = f->x_buf[i] * f->window[i];
    orc_mdct_fwd(f->mdct, f->mdct_buf, X);
}

void orc_amdct_synthesis(void *p, const double *X, double *x)
{
    orc_amdct_t *f = (orc_amdct_t *)p;
    const int F = f->frame_len;
    orc_mdct_inv(f->mdct, X, f->mdct_buf);                               /* llz_asmodel.c:446-461 */
    for (int i = 0; i < f->mdct_len; i++) {
        const double t = f->mdct_buf[i] * f->window[i];
        f->x_buf[i] = f->x_buf[i] + t;
    }
    for (int i = 0; i < F; i++) x[i] = f->x_buf[i];
    for (int i = 0; i < F; i++) f->x_buf[i] = f->x_buf[i + F];
    for (int i = 0; i < F; i++) f->x_buf[i + F] = 0;
}

void orc_amdct_free(void *p)
{
    orc_amdct_t *f = (orc_amdct_t *)p;
    if (!f) return;
    orc_mdct_free(f->mdct); free(f->x_buf); free(f->mdct_buf); free(f->window); free(f);
}

/* ------------------------------------------------------------------ batch drivers */

void orc_fir_batch_f32(const float *in, double *out, int channels, long n, const double *h, int flt_len)
{
    /* one orc_fir state machine per channel, a single frame of n samples (frame_len == init frame_len) */
    double *tmp = (double *)malloc(sizeof(double) * (size_t)n);
    for (int c = 0; c < channels; c++) {
        void *f = orc_fir_new_taps((int)n, h, flt_len);
        for (long i = 0; i < n; i++) tmp[i] = (double)in[(size_t)c * n + i];
        orc_fir_run(f, tmp, out + (size_t)c * n, (int)n);
        orc_fir_free(f);
    }
    free(tmp);
}

void orc_iir_cascade_batch_f32(const float *in, double *out, int channels, long n,
                               const double *coef, int stages)
{
    double *t0 = (double *)malloc(sizeof(double) * (size_t)n);
    double *t1 = (double *)malloc(sizeof(double) * (size_t)n);
    for (int c = 0; c < channels; c++) {
        for (long i = 0; i < n; i++) t0[i] = (double)in[(size_t)c * n + i];
        for (int s = 0; s < stages; s++) {
            const double *q = coef + 6 * s;
            void *f = orc_iir_new(2, q + 3, 2, q);
            orc_iir_run(f, t0, t1, (int)n);
            orc_iir_free(f);
            double *sw = t0; t0 = t1; t1 = sw;
        }
        memcpy(out + (size_t)c * n, t0, sizeof(double) * (size_t)n);
    }
    free(t0); free(t1);
}

long orc_rs_batch_f32(const float *in, double *out, int channels, long n_in, int L, int M,
                      double gain, int win)
{
    orc_rs_t *r = (orc_rs_t *)orc_rs_new(2, L, M, gain, win);
    if (!r) return -1;
    int Q = r->cols;
    long n_out = (n_in * L) / M;
    for (int c = 0; c < channels; c++) {
        const float *x = in + (size_t)c * n_in;
        double *y = out + (size_t)c * n_out;
        for (long i = 0; i < n_out; i++) {
            long pos = (i * M) / L;
            const double *g = r->mat + (size_t)(i % L) * Q;
            double acc = 0.0;
            for (int k = 0; k < Q; k++) {
                long idx = pos - k;
                double xv = (idx >= 0) ? (double)x[idx] : 0.0;
                acc += xv * g[k];
            }
            y[i] = acc * gain;
        }
    }
    orc_rs_free(r);
    return n_out;
}

long orc_rs_batch_i16(const short *in, short *out, int channels, long n_in, int L, int M,
                      double gain, int win)
{
    long n_out_total = -1;
    for (int c = 0; c < channels; c++) {
        orc_rs_t *r = (orc_rs_t *)orc_rs_new(2, L, M, gain, win);
        if (!r) return -1;
        long frames = n_in / r->num_in;
        int ob = 0;
        for (long f = 0; f < frames; f++)
            orc_rs_run(r, (const unsigned char *)(in + (size_t)c * n_in + (size_t)f * r->num_in),
                       2 * r->num_in,
                       (unsigned char *)(out + (size_t)c * (frames * r->num_out) + (size_t)f * r->num_out), &ob);
        n_out_total = frames * r->num_out;
        orc_rs_free(r);
    }
    return n_out_total;
}

static inline uint32_t fmix32(uint32_t u)
{
    u ^= u >> 16; u *= 0x85EBCA6Bu; u ^= u >> 13; u *= 0xC2B2AE35u; u ^= u >> 16;
    return u;
}

static inline uint32_t synth_u32(uint32_t seed, uint32_t c, uint32_t n)
{
    return fmix32(seed ^ (c * 0x9E3779B9u) ^ (n * 0x85EBCA6Bu));
}

void orc_synth_f32(float *dst, int channels, long n, unsigned seed, int chan0)
{
    for (int c = 0; c < channels; c++)
        for (long i = 0; i < n; i++) {
            uint32_t u = synth_u32(seed, (uint32_t)(c + chan0), (uint32_t)i);
            dst[(size_t)c * n + i] = (float)(u >> 8) * (1.0f / 8388608.0f) - 1.0f;
        }
}

void orc_synth_i16(short *dst, int channels, long n, unsigned seed, int chan0)
{
    for (int c = 0; c < channels; c++)
        for (long i = 0; i < n; i++) {
            uint32_t u = synth_u32(seed, (uint32_t)(c + chan0), (uint32_t)i);
            dst[(size_t)c * n + i] = (short)((int32_t)(u >> 17) - 16384);
        }
}


/* ---- PCM ingest / egress and the WAV header ------------------------------------------------- */

/* planar float = interleaved int16 * scale: the data order of a WAV data chunk is sample-major, channels inside a sample
 * (what example/llz_resample/main.c:96-114 reads for one channel) */
void orc_pcm_deinterleave_i16_f32(const short *in, float *out, int channels, long n, float scale)
{
    for (long i = 0; i < n; i++)
        for (int c = 0; c < channels; c++) out[(size_t)c * n + i] = (float)in[(size_t)i * channels + c] * scale;
}

/* the reference's float -> int16 rule: clamp to [-32768, 32767], then the C conversion toward zero (llz_resample.c:596-601) */
void orc_pcm_interleave_f32_i16(const float *in, short *out, int channels, long n, float scale)
{
    for (long i = 0; i < n; i++)
        for (int c = 0; c < channels; c++) {
            float y = in[(size_t)c * n + i] * scale;
            if (y > 32767) y = 32767;
            if (y < -32768) y = -32768;
            out[(size_t)i * channels + c] = (short)y;
        }
}

static unsigned long orc_le(const unsigned char *p, int bytes)
{
    unsigned long v = 0;
    for (int i = bytes - 1; i >= 0; i--) v = (v << 8) | p[i];
    return v;
}

/* The chunk walk of llz_wavfmt_readheader (libllzaudio/llz_wavfmt.c:82-159) on a memory image: "RIFF" size "WAVE", chunks
 * skipped up to "fmt ", PCM format, channels, rate, bits -> bytes per sample (rounded up), block_align recomputed as
 * bytes_per_sample * channels (:136), the rest of the fmt chunk skipped, chunks skipped up to "data", frames = data bytes /
 * block_align (:155).  Returns 0, or -1 where the reference prints and exits / would read past the file.
 * out: format, channels, samplerate, bytes_per_sample, block_align, frames, data_offset */
int orc_wav_parse(const unsigned char *b, long len, long *out)
{
    long pos = 0;
    if (len < 12 || memcmp(b, "RIFF", 4) != 0 || memcmp(b + 8, "WAVE", 4) != 0) return -1;
    pos = 12;
    for (;;) {                                                   /* llz_wavfmt.c:108-114 (the id is re-read after a skip) */
        if (pos + 8 > len) return -1;
        if (memcmp(b + pos, "fmt ", 4) == 0) break;
        pos += 8 + (long)orc_le(b + pos + 4, 4);
    }
    long x_size = (long)orc_le(b + pos + 4, 4);
    pos += 8;
    if (x_size < 16 || pos + x_size > len) return -1;
    const unsigned long format = orc_le(b + pos, 2), channels = orc_le(b + pos + 2, 2);
    const unsigned long rate = orc_le(b + pos + 4, 4), bits = orc_le(b + pos + 14, 2);
    if (format != 1 || channels == 0) return -1;                 /* WAVE_FORMAT_PCM only (:118-121) */
    const unsigned long bps = (bits + 7) / 8;
    if (bps == 0) return -1;
    pos += x_size;
    for (;;) {                                                   /* :146-152 */
        if (pos + 8 > len) return -1;
        if (memcmp(b + pos, "data", 4) == 0) break;
        pos += 8 + (long)orc_le(b + pos + 4, 4);
    }
    const unsigned long data_size = orc_le(b + pos + 4, 4);
    out[0] = (long)format; out[1] = (long)channels; out[2] = (long)rate; out[3] = (long)bps;
    out[4] = (long)(bps * channels); out[5] = (long)(data_size / (bps * channels)); out[6] = pos + 8;
    return 0;
}

/* the 44-byte header of llz_wavfmt_writeheader (:185-213) */
void orc_wav_header(unsigned char *h, int channels, long samplerate, int bytes_per_sample, long frames)
{
    const unsigned long block = (unsigned long)channels * bytes_per_sample, data = (unsigned long)frames * block;
    memcpy(h, "RIFF", 4);
    unsigned long v = data + 36;
    for (int i = 0; i < 4; i++) h[4 + i] = (unsigned char)(v >> (8 * i));
    memcpy(h + 8, "WAVEfmt ", 8);
    v = 16;
    for (int i = 0; i < 4; i++) h[16 + i] = (unsigned char)(v >> (8 * i));
    h[20] = 1; h[21] = 0;
    h[22] = (unsigned char)channels; h[23] = (unsigned char)(channels >> 8);
    for (int i = 0; i < 4; i++) h[24 + i] = (unsigned char)((unsigned long)samplerate >> (8 * i));
    v = (unsigned long)channels * (unsigned long)samplerate * bytes_per_sample;
    for (int i = 0; i < 4; i++) h[28 + i] = (unsigned char)(v >> (8 * i));
    h[32] = (unsigned char)block; h[33] = (unsigned char)(block >> 8);
    h[34] = (unsigned char)(bytes_per_sample * 8); h[35] = 0;
    memcpy(h + 36, "data", 4);
    for (int i = 0; i < 4; i++) h[40 + i] = (unsigned char)(data >> (8 * i));
}
