/*
 * llz_design.c -- host-side filter design of libllzfilter_hip.so: windows, tap-count estimators, windowed-sinc
 * low/high/band-pass/band-stop taps and the host dot product.  Exports the reference's symbols
 * (reference libllzfilter/llz_fir.h:64-94).  Setup-time code: runs once per handle on the CPU, in double, with
 * the reference's operation order so that the taps -- which are data for every kernel -- come out identical
 * (SURVEY.md H5).  Compiled with -ffp-contract=off.
 */
#include <math.h>
#include <stdlib.h>
#include "../../../include/llz_fir.h"
#include "llz_host.h"

/* ---- windows: llz_fir.c:61-158 ---- */

int llz_hamming(double *w, const int N)
{
    int lo = 0, hi = N - 1;
    while (lo <= hi) {
        double v = 0.54 - 0.46 * cos(2 * M_PI * lo / (N - 1));
        w[lo++] = v;
        w[hi--] = v;
    }
    return N;
}

int llz_blackman(double *w, const int N)
{
    int lo = 0, hi = N - 1;
    while (lo <= hi) {
        double v = 0.42 - 0.5 * cos(2 * M_PI * lo / (N - 1)) + 0.08 * cos(4 * M_PI * lo / (N - 1));
        w[lo++] = v;
        w[hi--] = v;
    }
    return N;
}

/* I0(x): power series, terminated when the term falls below 1e-16 of the running sum (llz_fir.c:85-103) */
static double i0_series(double x)
{
    const double h = 0.5 * x;
    double total = 1.0, factor = 1.0, term = 1.0;
    for (int k = 1; term > total * 1E-16; k++) {
        factor = factor * (h / k);
        term = factor * factor;
        total = total + term;
    }
    return total;
}

int llz_kaiser_beta(double *w, const int N, const double beta)
{
    const double norm = i0_series(beta);
    for (int i = 0; i < N; i++) {
        const double x = (2. * i / (N - 1)) - 1;
        w[i] = i0_series(beta * sqrt(1. - x * x)) / norm;
    }
    return N;
}

int llz_kaiser(double *w, const int N)
{
    return llz_kaiser_beta(w, N, 8.96);       /* llz_fir.c:125 */
}

double llz_kaiser_atten2beta(double atten)
{
    if (atten <= 21.) return 0.;
    if (atten < 50.) return 0.5842 * pow(atten - 21., 0.4) + 0.07886 * (atten - 21.);
    return 0.1102 * (atten - 8.7);
}

/* ---- estimators: llz_fir.c:173-193 ---- */

int llz_hamming_cof_num(double ftrans)  { return (int)(6.2 / ftrans); }
int llz_blackman_cof_num(double ftrans) { return (int)(6.6 / ftrans); }

int llz_kaiser_cof_num(double ftrans, double atten)
{
    return atten <= 21. ? (int)((0.9222 * 2.) / ftrans)
                        : (int)(((atten - 7.95) * 2.) / (14.36 * ftrans));
}

/* ---- taps: llz_fir.c:39-59, 201-393 ---- */

static double sinc_norm(double x)
{
    if (x == 0.0) return 1.0;
    if (x == floor(x)) return 0.0;            /* exact zero crossings */
    return sin(M_PI * fmod(x, 2.0)) / (M_PI * x);
}

static int fill_window(double *w, int n, win_t win)
{
    switch (win) {
    case HAMMING:  llz_hamming(w, n);  return 0;
    case BLACKMAN: llz_blackman(w, n); return 0;
    case KAISER:   llz_kaiser(w, n);   return 0;
    }
    return -1;
}

int llz_host_design(int kind, double **out, int n, double fc1, double fc2, win_t win)
{
    if (n < 1) return -1;
    if (kind != LLZ_KIND_LPF && (n & 1) == 0) n += 1;          /* llz_fir.c:305-307 */
    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    double *h = (double *)malloc(sizeof(double) * (size_t)n);
    if (!w || !h || fill_window(w, n, win) != 0) {
        free(w); free(h);
        return -1;
    }
    if (kind == LLZ_KIND_LPF) {
        const double centre = (double)(n - 1) / 2;             /* llz_fir.c:206: may be x.5 for even n */
        for (int i = 0, j = n - 1; i <= centre; i++, j--)
            h[i] = h[j] = fc1 * sinc_norm(fc1 * (i - centre)) * w[i];
    } else {
        const int centre = (n - 1) / 2;
        for (int i = 0, j = n - 1; i <= centre; i++, j--) {
            const int d = i - centre;
            double v;
            switch (kind) {
            case LLZ_KIND_HPF: v = -fc1 * sinc_norm(fc1 * d) * w[i]; break;                               /* :226 */
            case LLZ_KIND_BPF: v = (fc2 * sinc_norm(fc2 * d) - fc1 * sinc_norm(fc1 * d)) * w[i]; break;   /* :244 */
            default:           v = -(fc2 * sinc_norm(fc2 * d) - fc1 * sinc_norm(fc1 * d)) * w[i]; break;  /* :262 */
            }
            h[i] = h[j] = v;
        }
        h[centre] = kind == LLZ_KIND_HPF ? 1 - fc1 : kind == LLZ_KIND_BPF ? fc2 - fc1 : 1 - (fc2 - fc1);
    }
    free(w);
    *out = h;
    return n;
}

int llz_fir_lpf_cof(double **h, int N, double fc, win_t win_type)
{
    return llz_host_design(LLZ_KIND_LPF, h, N, fc, 0.0, win_type);
}

int llz_fir_hpf_cof(double **h, int N, double fc, win_t win_type)
{
    return llz_host_design(LLZ_KIND_HPF, h, N, fc, 0.0, win_type);
}

int llz_fir_bandpass_cof(double **h, int N, double fc1, double fc2, win_t win_type)
{
    return llz_host_design(LLZ_KIND_BPF, h, N, fc1, fc2, win_type);
}

int llz_fir_bandstop_cof(double **h, int N, double fc1, double fc2, win_t win_type)
{
    return llz_host_design(LLZ_KIND_BSF, h, N, fc1, fc2, win_type);
}

double llz_conv(const double *x, const double *h, int h_len)
{
    /* llz_fir.c:411-426; a host utility the reference exports, not a kernel path */
    double y = 0.0;
    for (int i = 0; i < h_len; i++)
        y += h[i] * x[-i];
    return y;
}
