// corr.hip -- auto / cross correlation kernels (reference libllzfilter/llz_corr.c:38-177).
//
//  k_corr_exact_f64    one lane per lag, the reference's running sum in its order (rounded multiply, rounded add):
//                      bit-identical to llz_autocorr / llz_crosscorr; also the three sums of llz_corr_cof.
//  k_autocorr_mc_f32   one workgroup per frame; the frame is walked in 8192-sample chunks staged in LDS, every lane keeps
//                      one accumulator per lag of the current group of 32 lags; HBM sees each sample once.
//  k_acf_pack / k_acf_power / k_acf_extract
//                      the pointwise steps of the FFT form around the batched float32 FFT of fft.hip:
//                      zero-padded real -> complex, |X|^2 of the FIRST n bins (the reference's definition), 2*Re.
#include "common.hpp"

namespace {

__global__ void __launch_bounds__(64)
k_corr_exact_f64(const double *__restrict__ x, const double *__restrict__ y, int n, int p, double *__restrict__ r)
{
#pragma clang fp contract(off)
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k > p) return;
    double acc = 0.0;
    for (int i = 0; i + k < n; i++) {
        const double prod = x[i] * y[i + k];
        acc = acc + prod;
    }
    r[k] = acc;
}

constexpr int AC_THREADS = 256;
constexpr int AC_CHUNK = 8192;          // samples staged per pass (+ 255 halo)
constexpr int AC_LG = 32;               // lags per accumulator group

__global__ void __launch_bounds__(AC_THREADS)
k_autocorr_mc_f32(const float *__restrict__ x, float *__restrict__ r, int n, int p)
{
    __shared__ float xs[AC_CHUNK + 256];
    __shared__ float red[AC_THREADS / 64][AC_LG];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *row = x + (size_t)f * n;
    for (int g0 = 0; g0 <= p; g0 += AC_LG) {
        float acc[AC_LG];
#pragma unroll
        for (int k = 0; k < AC_LG; k++) acc[k] = 0.f;
        for (int c0 = 0; c0 < n; c0 += AC_CHUNK) {
            const int len = min(AC_CHUNK, n - c0);              // positions i in [c0, c0+len)
            const int span = min(len + g0 + AC_LG, n - c0);     // samples needed: up to i + k
            __syncthreads();
            for (int i = tid; i < span; i += AC_THREADS) xs[i] = row[c0 + i];
            for (int i = span + tid; i < len + g0 + AC_LG; i += AC_THREADS) xs[i] = 0.f;   // beyond the frame: zero terms
            __syncthreads();
            for (int i = tid; i < len; i += AC_THREADS) {
                const float xi = xs[i];
#pragma unroll
                for (int k = 0; k < AC_LG; k++) acc[k] = __builtin_fmaf(xi, xs[i + g0 + k], acc[k]);
            }
        }
        // reduce over the lanes of each wave, then over the waves
#pragma unroll
        for (int k = 0; k < AC_LG; k++) {
            float v = acc[k];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
            if (lane == 0) red[wave][k] = v;
        }
        __syncthreads();
        if (tid < AC_LG && g0 + tid <= p) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < AC_THREADS / 64; w++) v += red[w][tid];
            r[(size_t)f * (p + 1) + g0 + tid] = v;
        }
    }
}

// zero-padded real frame -> interleaved complex of length F
__global__ void __launch_bounds__(256)
k_acf_pack(const float *__restrict__ x, float2 *__restrict__ z, int n, int F, long total)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;                  // flat index over frames x F
    if (e >= total) return;
    const long f = e / F;
    const int i = (int)(e - f * F);
    z[e] = make_float2(i < n ? x[f * n + i] : 0.f, 0.f);
}

// power spectrum of the first n bins, zero elsewhere (llz_corr.c:165-170)
__global__ void __launch_bounds__(256)
k_acf_power(float2 *__restrict__ z, int n, int F, long total)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int i = (int)(e % F);
    const float2 v = z[e];
    z[e] = make_float2(i < n ? __builtin_fmaf(v.x, v.x, v.y * v.y) : 0.f, 0.f);
}

__global__ void __launch_bounds__(256)
k_acf_extract(const float2 *__restrict__ z, float *__restrict__ r, int p, int F, long total)
{
    const long e = (long)blockIdx.x * 256 + threadIdx.x;                  // flat index over frames x (p+1)
    if (e >= total) return;
    const long f = e / (p + 1);
    const int k = (int)(e - f * (p + 1));
    r[e] = z[f * F + k].x * 2.f;                                          // llz_corr.c:173
}

} // namespace

extern "C" int llzs_corr_exact_f64(const double *x, const double *y, int n, int p, double *r, void *stream)
{
    if (!x || !y || !r || n < 1 || p < 0) {
        llzs_set_error("corr_exact_f64: bad arguments");
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_corr_exact_f64, dim3((unsigned)(p / 64 + 1)), dim3(64), 0, as_stream(stream), x, y, n, p, r);
    LLZ_LAUNCH_CHECK("k_corr_exact_f64");
    return LLZ_OK;
}

extern "C" int llzs_autocorr_mc_f32(const float *x, float *r, int frames, int n, int p, void *stream)
{
    if (!x || !r || frames < 1 || n < 1 || p < 0 || p >= n || p > 255) {
        llzs_set_error("autocorr_mc_f32: bad arguments (frames=%d n=%d p=%d; p < n, p <= 255)", frames, n, p);
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_autocorr_mc_f32, dim3((unsigned)frames), dim3(AC_THREADS), 0, as_stream(stream), x, r, n, p);
    LLZ_LAUNCH_CHECK("k_autocorr_mc_f32");
    return LLZ_OK;
}

extern "C" int llzs_acf_pack(const float *x, float *z, int frames, int n, int F, void *stream)
{
    const long total = (long)frames * F;
    hipLaunchKernelGGL(k_acf_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), x,
                       reinterpret_cast<float2 *>(z), n, F, total);
    LLZ_LAUNCH_CHECK("k_acf_pack");
    return LLZ_OK;
}

extern "C" int llzs_acf_power(float *z, int frames, int n, int F, void *stream)
{
    const long total = (long)frames * F;
    hipLaunchKernelGGL(k_acf_power, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<float2 *>(z), n, F, total);
    LLZ_LAUNCH_CHECK("k_acf_power");
    return LLZ_OK;
}

extern "C" int llzs_acf_extract(const float *z, float *r, int frames, int p, int F, void *stream)
{
    const long total = (long)frames * (p + 1);
    hipLaunchKernelGGL(k_acf_extract, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2 *>(z), r, p, F, total);
    LLZ_LAUNCH_CHECK("k_acf_extract");
    return LLZ_OK;
}
