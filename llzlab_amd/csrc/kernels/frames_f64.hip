// frames_f64.hip -- the framing around the single-channel analysis / synthesis transforms, on the device in double
// (SURVEY.md 8(f) rank 3 and the frames of rank 4; reference libllzfilter/llz_asmodel.c:177-203, :274-309, :365-383,
// :440-462).  A handle's running buffers live in device memory; a call moves one frame of samples (or one spectrum) in
// and one out, everything between the two copies is a kernel.
//
// One lane per element, every product and every sum rounded separately (no contraction), which is what the reference's
// `a += b * w` compiles to on its x86-64 baseline build: results are bit-identical to the CPU library.
//
//  k_frame_slide_window   analysis: held' = (held << hop) ++ fresh;  dst = held' * window  (complex with zero imaginary
//                         part for the FFT frames, real for the MDCT frames)
//  k_spectrum_split       bins 0..N/2 of an interleaved spectrum -> a real plane and an imaginary plane
//  k_spectrum_mirror      the two planes -> the full interleaved spectrum of a real signal (upper half = conjugate mirror)
//  k_frame_overlap_add    synthesis: acc' = acc + src * window; the first hop values leave (scaled), the rest slide down,
//                         zeros enter at the top
//  k_scale_4_over_n       v -> (v * 4) / n, two roundings (tail of the defining-sum IMDCT, llz_mdct.c:219-220)
#include "common.hpp"

namespace {

__global__ void __launch_bounds__(256)
k_frame_slide_window(const double *__restrict__ fresh, const double *__restrict__ held, double *__restrict__ held_next,
                     const double *__restrict__ window, double *__restrict__ dst, int N, int hop, int as_complex)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int kept = N - hop;
    const double v = i < kept ? held[i + hop] : fresh[i - kept];
    held_next[i] = v;
    const double p = v * window[i];
    if (as_complex) {
        dst[2 * i] = p;
        dst[2 * i + 1] = 0;
    } else {
        dst[i] = p;
    }
}

__global__ void __launch_bounds__(256)
k_spectrum_split(const double *__restrict__ z, double *__restrict__ planes, int bins)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= bins) return;
    planes[i] = z[2 * i];
    planes[bins + i] = z[2 * i + 1];
}

__global__ void __launch_bounds__(256)
k_spectrum_mirror(const double *__restrict__ planes, double *__restrict__ z, int N)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int bins = (N >> 1) + 1;
    if (i < bins) {
        z[2 * i] = planes[i];
        z[2 * i + 1] = planes[bins + i];
    } else {
        z[2 * i] = planes[N - i];
        z[2 * i + 1] = -planes[bins + N - i];
    }
}

__global__ void __launch_bounds__(256)
k_frame_overlap_add(const double *__restrict__ src, int src_stride, const double *__restrict__ window,
                    const double *__restrict__ acc, double *__restrict__ acc_next, double *__restrict__ leaving, int N,
                    int hop, double scale)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double t = src[(size_t)i * src_stride] * window[i];
    const double a = acc[i] + t;
    if (i < hop) leaving[i] = scale * a;
    else acc_next[i - hop] = a;
    if (i >= N - hop) acc_next[i] = 0;
}

__global__ void __launch_bounds__(256)
k_scale_4_over_n(double *__restrict__ v, int n, double divisor)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double q = v[i] * 4;
    v[i] = q / divisor;
}

inline dim3 lanes(int n) { return dim3((unsigned)((n + 255) / 256)); }

} // namespace

extern "C" int llzs_frame_slide_window_f64(const double *fresh, const double *held, double *held_next,
                                           const double *window, double *dst, int N, int hop, int as_complex,
                                           void *stream)
{
    if (!fresh || !held || !held_next || held == held_next || !window || !dst || N < 1 || hop < 1 || hop > N) {
        llzs_set_error("frame_slide_window_f64: bad arguments");
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_frame_slide_window, lanes(N), dim3(256), 0, as_stream(stream), fresh, held, held_next, window,
                       dst, N, hop, as_complex);
    LLZ_LAUNCH_CHECK("k_frame_slide_window");
    return LLZ_OK;
}

extern "C" int llzs_spectrum_split_f64(const double *z, double *planes, int bins, void *stream)
{
    if (!z || !planes || bins < 1) {
        llzs_set_error("spectrum_split_f64: bad arguments");
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_spectrum_split, lanes(bins), dim3(256), 0, as_stream(stream), z, planes, bins);
    LLZ_LAUNCH_CHECK("k_spectrum_split");
    return LLZ_OK;
}

extern "C" int llzs_spectrum_mirror_f64(const double *planes, double *z, int N, void *stream)
{
    if (!z || !planes || N < 2) {
        llzs_set_error("spectrum_mirror_f64: bad arguments");
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_spectrum_mirror, lanes(N), dim3(256), 0, as_stream(stream), planes, z, N);
    LLZ_LAUNCH_CHECK("k_spectrum_mirror");
    return LLZ_OK;
}

extern "C" int llzs_frame_overlap_add_f64(const double *src, int src_stride, const double *window, const double *acc,
                                          double *acc_next, double *leaving, int N, int hop, double scale, void *stream)
{
    if (!src || src_stride < 1 || !window || !acc || !acc_next || acc == acc_next || !leaving || N < 1 || hop < 1 ||
        hop > N) {
        llzs_set_error("frame_overlap_add_f64: bad arguments");
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_frame_overlap_add, lanes(N), dim3(256), 0, as_stream(stream), src, src_stride, window, acc,
                       acc_next, leaving, N, hop, scale);
    LLZ_LAUNCH_CHECK("k_frame_overlap_add");
    return LLZ_OK;
}

extern "C" int llzs_scale_4_over_n_f64(double *v, int n, double divisor, void *stream)
{
    if (!v || n < 1 || divisor == 0) {
        llzs_set_error("scale_4_over_n_f64: bad arguments");
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_scale_4_over_n, lanes(n), dim3(256), 0, as_stream(stream), v, n, divisor);
    LLZ_LAUNCH_CHECK("k_scale_4_over_n");
    return LLZ_OK;
}
