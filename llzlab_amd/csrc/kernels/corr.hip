// corr.hip -- auto / cross correlation kernels (reference libllzfilter/llz_corr.c:38-177).
//
//  k_corr_exact_f64    one lane per lag, the reference's running sum in its order (rounded multiply, rounded add):
//                      bit-identical to llz_autocorr / llz_crosscorr; also the three sums of llz_corr_cof.
//  k_autocorr_mc_f32   one wave per frame, 512-sample chunks in the wave's private LDS, register sliding window
//                      (64 FMAs per two ds_read_b128), partial sums in registers, one DPP reduction per frame.
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

// Direct autocorrelation of many frames.  A WAVE owns a frame (frames are dealt round robin to the waves of a
// persistent grid) and walks it in chunks of 512 samples: the chunk plus p samples of look-ahead is staged in the wave's
// private LDS (coalesced dword loads, no workgroup barrier), lane l keeps x[8l .. 8l+7] in registers and slides a
// 15-sample window over the lags, 8 lags at a time: 64 FMAs per two ds_read_b128 -- the register-window scheme of
// fir_td.hip with the frame itself in the role of the taps.  Per-lane partial sums live in registers for the whole
// frame and are reduced across the wave once per frame with DPP adds.
constexpr int AC_WAVES = 4;
constexpr int AC_CHUNK = 512;                 // samples per wave and step: 8 per lane
constexpr int AC_MAXLAG = 256;                // p <= 255
constexpr int AC_LDS = (AC_CHUNK + AC_MAXLAG + 16) + ((AC_CHUNK + AC_MAXLAG + 16) >> 3) * 4;   // padded image

__device__ __forceinline__ int ac_phys(int p) { return p + ((p >> 3) << 2); }

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

template <int NG>                               // lag groups of 8 kept in registers: lags 0 .. 8*NG-1
__global__ void __launch_bounds__(64 * AC_WAVES)
k_autocorr_mc_f32(const float *__restrict__ x, float *__restrict__ r, int frames, int n, int p, int lag0)
{
    __shared__ __attribute__((aligned(16))) float lds_ac[AC_WAVES][AC_LDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *xs = lds_ac[wave];
    const long waves_total = (long)gridDim.x * AC_WAVES;
    const int look = lag0 + 8 * NG;                            // look-ahead samples a chunk needs behind its end
                                                               // (this launch does lags lag0 .. lag0 + 8*NG - 1)
    for (long f = (long)blockIdx.x * AC_WAVES + wave; f < frames; f += waves_total) {
        const float *row = x + (size_t)f * n;
        float acc[8 * NG];
#pragma unroll
        for (int k = 0; k < 8 * NG; k++) acc[k] = 0.f;
        for (int c0 = 0; c0 < n; c0 += AC_CHUNK) {
            // stage x[c0 .. c0 + 512 + look): zeros behind the end of the frame make those products vanish
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the previous step's reads are done
            for (int i = lane; i < AC_CHUNK + look; i += 64) {
                const int idx = c0 + i;
                xs[ac_phys(i)] = idx < n ? row[idx] : 0.f;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int p0 = lane * 8;
            float xl[8], wa[8], wb[8];
            auto load8 = [&](float (&w)[8], int q) {
                const float4 a = *reinterpret_cast<const float4 *>(&xs[ac_phys(q)]);
                const float4 b = *reinterpret_cast<const float4 *>(&xs[ac_phys(q + 4)]);
                w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
                w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
            };
            load8(xl, p0);
            load8(wa, p0 + lag0);
            // lag group g: lags 8g .. 8g+7 need x[p0 + 8g .. p0 + 8g + 14] = (wa | wb) with wb = the next 8 samples
#pragma unroll
            for (int g = 0; g < NG; g++) {
                load8(wb, p0 + lag0 + 8 * g + 8);
#pragma unroll
                for (int kk = 0; kk < 8; kk++)
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int slot = j + kk;                           // 0..14
                        acc[8 * g + kk] = __builtin_fmaf(xl[j], slot < 8 ? wa[slot] : wb[slot - 8], acc[8 * g + kk]);
                    }
#pragma unroll
                for (int j = 0; j < 8; j++) wa[j] = wb[j];
            }
        }
#pragma unroll
        for (int k = 0; k < 8 * NG; k++) {
            const float v = wave_sum(acc[k]);
            if (lane == 0 && lag0 + k <= p) r[(size_t)f * (p + 1) + lag0 + k] = v;
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
    long blocks = ((long)frames + AC_WAVES - 1) / AC_WAVES;
    if (blocks > 256L * 4) blocks = 256L * 4;                       // persistent: frames dealt round robin to the waves
    // lags are done 64 per launch (8 groups of 8 accumulators per lane); p > 63 re-reads the frames per block of lags
#define LLZ_AC_LAUNCH(NG, LAG0)                                                                                   \
    hipLaunchKernelGGL(k_autocorr_mc_f32<NG>, dim3((unsigned)blocks), dim3(64 * AC_WAVES), 0, as_stream(stream), x, r, \
                       frames, n, p, LAG0)
    for (int lag0 = 0; lag0 <= p; lag0 += 64) {
        const int ng = (((p - lag0) < 63 ? (p - lag0) : 63) + 8) / 8;
        if (ng <= 1) LLZ_AC_LAUNCH(1, lag0);
        else if (ng <= 2) LLZ_AC_LAUNCH(2, lag0);
        else if (ng <= 3) LLZ_AC_LAUNCH(3, lag0);
        else if (ng <= 4) LLZ_AC_LAUNCH(4, lag0);
        else if (ng <= 5) LLZ_AC_LAUNCH(5, lag0);
        else LLZ_AC_LAUNCH(8, lag0);
    }
#undef LLZ_AC_LAUNCH
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
