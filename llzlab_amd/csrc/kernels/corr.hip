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

// All K per-lane partial sums of a wave at once by recursive halving: in the step with distance d a lane keeps one accumulator of a
// pair and hands the other to lane ^ d, which keeps that one -- the number of live sums halves with every step (17 -> 9 -> 5 -> 3
// -> 2 -> 1 -> 1: 21 exchanges instead of 17 x 6).  Returns the total of sum number (six bits of the lane, reversed) -- for
// K <= 64 every sum ends in exactly one lane; *which says which.
template <int K>
__device__ __forceinline__ float wave_sums(float (&acc)[K], int lane, int *which)
{
    static_assert(K <= 64, "one sum per lane at most");
    int cnt = K;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const bool upper = (lane & d) != 0;
        const int half = (cnt + 1) / 2;
#pragma unroll
        for (int i = 0; i < (K + 1) / 2; i++) {
            if (i < half) {
                const float a = acc[2 * i], b = 2 * i + 1 < cnt ? acc[2 * i + 1] : 0.f;
                acc[i] = (upper ? b : a) + __shfl_xor(upper ? a : b, d, 64);
            }
        }
        cnt = half;
    }
    *which = (int)(__brev((unsigned)lane) >> 26);
    return acc[0];
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
        int k;
        const float v = wave_sums(acc, lane, &k);
        if (k < 8 * NG && lag0 + k <= p) r[(size_t)f * (p + 1) + lag0 + k] = v;
    }
}

// Direct autocorrelation for SHORT lag ranges (p <= 8 NL <= 32: the LPC orders) without LDS: lane l keeps x[8 l .. 8 l + 7] of a
// chunk in registers and gets the samples it slides over from lanes l + 1 .. l + NL through the wave shuffle, so a chunk is
// 8 (64 - NL) samples (the last NL lanes only look ahead: their own products are formed by the next chunk, where they are the
// first lanes); the next chunk's samples are requested before the current one is worked on.  One 16-byte-pair load and
// 8 (p + 1) FMAs per lane and chunk; the per-lane partial sums are reduced across the wave once per frame.  (The LDS form
// above -- staging, two waits and the window reads per 512 samples, nothing in flight meanwhile -- measured 0.93 ms for 2^18
// frames of 1024 at p = 16; this one is bound by the reduction and the FMAs.)
typedef float ac_f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) ac_x4 { ac_f32x4 v; };     // a 16-byte load at any 4-byte address

template <int NL>
__global__ void __launch_bounds__(64 * AC_WAVES)
k_autocorr_reg_f32(const float *__restrict__ x, float *__restrict__ r, int frames, int n, int p)
{
    constexpr int NLAG = 8 * NL + 1, STEP = 8 * (64 - NL);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long waves_total = (long)gridDim.x * AC_WAVES;
    const bool active = lane < 64 - NL;
    for (long f = (long)blockIdx.x * AC_WAVES + wave; f < frames; f += waves_total) {
        const float *row = x + (size_t)f * n;
        auto fetch = [&](int c0, float (&v)[8]) {
            const int i0 = c0 + 8 * lane;
            if (i0 + 8 <= n) {
                const ac_f32x4 a = reinterpret_cast<const ac_x4 *>(row + i0)->v, b = reinterpret_cast<const ac_x4 *>(row + i0 + 4)->v;
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
                v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = i0 + j < n ? row[i0 + j] : 0.f;   // zeros behind the frame: those products vanish
            }
        };
        float acc[NLAG];
#pragma unroll
        for (int k = 0; k < NLAG; k++) acc[k] = 0.f;
        float cur[8], nxt[8];
        fetch(0, cur);
        for (int c0 = 0; c0 < n; c0 += STEP) {
            if (c0 + STEP < n) fetch(c0 + STEP, nxt);
            // s = the lane's own samples followed by those of lanes l + 1 .. l + NL
            float s[8 * (NL + 1)];
#pragma unroll
            for (int j = 0; j < 8; j++) s[j] = cur[j];
#pragma unroll
            for (int h = 1; h <= NL; h++)
#pragma unroll
                for (int j = 0; j < 8; j++) s[8 * h + j] = __shfl_down(s[8 * (h - 1) + j], 1, 64);
            float xa[8];
#pragma unroll
            for (int j = 0; j < 8; j++) xa[j] = active ? cur[j] : 0.f;
#pragma unroll
            for (int k = 0; k < NLAG; k++)
#pragma unroll
                for (int j = 0; j < 8; j++) acc[k] = __builtin_fmaf(xa[j], s[j + k], acc[k]);
#pragma unroll
            for (int j = 0; j < 8; j++) cur[j] = nxt[j];
        }
        int k;
        const float v = wave_sums(acc, lane, &k);
        if (k <= p) r[(size_t)f * (p + 1) + k] = v;
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
    if (p <= 32 && llzs_tune(LLZS_TUNE_ACF_LDS) < 1) {              // short lag ranges: the register form
        const int nl = p <= 8 ? 1 : (p + 7) / 8;
#define LLZ_AC_REG(NLV)                                                                                            \
    hipLaunchKernelGGL(k_autocorr_reg_f32<NLV>, dim3((unsigned)blocks), dim3(64 * AC_WAVES), 0, as_stream(stream), x, r, \
                       frames, n, p)
        if (nl == 1) LLZ_AC_REG(1);
        else if (nl == 2) LLZ_AC_REG(2);
        else if (nl == 3) LLZ_AC_REG(3);
        else LLZ_AC_REG(4);
#undef LLZ_AC_REG
        LLZ_LAUNCH_CHECK("k_autocorr_reg_f32");
        return LLZ_OK;
    }
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
