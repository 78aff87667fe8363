// resample.hip -- K3: rational L/M resampler, decimator and interpolator for gfx950.
//
// Replaces the hot loops of reference libllzfilter/llz_resample.c:583-603 (llz_resample), :457-483 (llz_decimate)
// and :515-536 (llz_interp).  Indexing is the reference's:  y[i] = gain * sum_{k<Q} x[(i*M)/L - k] * g[i mod L][k].
//
//  k_resample<float,float>    float32 in/out, float accumulate (FMA), no clamp: the batch extension.
//  k_resample<short,double>   int16 in/out: double accumulate in ascending k with a rounded multiply and a rounded
//                             add per tap (no contraction), * gain, clamp to [-32768,32767], truncate toward zero --
//                             bit-exact with the reference for every sample.
//
// One workgroup = one channel x 256 consecutive outputs. The input span those outputs touch ((256*M)/L + Q samples)
// is staged once in LDS with coalesced reads, so HBM sees each input sample about once (+Q/span halo from L2).
// Tap rows come through the vector L1 (one row for L = 1: every lane reads the same address).
#include "common.hpp"

namespace {

constexpr int RS_THREADS = 256;

template <typename T>
struct rs_traits;
template <>
struct rs_traits<float> {
    typedef float acc_t;
    typedef float tap_t;
};
template <>
struct rs_traits<short> {
    typedef double acc_t;
    typedef double tap_t;
};

__device__ __forceinline__ float rs_finish(float acc, float gain, float *) { return acc * gain; }

__device__ __forceinline__ short rs_finish(double acc, double gain, short *)
{
#pragma clang fp contract(off)
    double y = acc * gain;                        // llz_resample.c:594
    if (y > 32767) y = 32767;
    if (y < -32768) y = -32768;
    return (short)y;                              // C conversion: toward zero (v_cvt_i32_f64 truncates)
}

template <typename T>
__global__ void __launch_bounds__(RS_THREADS)
k_resample(const T *__restrict__ in, T *__restrict__ out, const T *__restrict__ hist,
           const typename rs_traits<T>::tap_t *__restrict__ g, long n_in, long n_out, long in_pitch,
           long out_pitch, int L, int M, int Q, typename rs_traits<T>::acc_t gain, long long i0, long long in0,
           int span_max)
{
#pragma clang fp contract(off)
    typedef typename rs_traits<T>::acc_t acc_t;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *xs = reinterpret_cast<T *>(smem_raw);
    const int c = blockIdx.y;
    const long o0 = (long)blockIdx.x * RS_THREADS;                 // first output (local) of this tile
    const long olast = min(o0 + RS_THREADS, n_out) - 1;
    // local input index of the newest sample an output needs: ((i0+i)*M)/L - in0
    const long pos_first = (long)(((i0 + o0) * M) / L - in0);
    const long pos_last = (long)(((i0 + olast) * M) / L - in0);
    const long base = pos_first - (Q - 1);                         // oldest local input index staged
    const int span = (int)(pos_last - base + 1);                   // <= span_max by construction
    const T *row = in + (size_t)c * in_pitch;
    const T *hrow = hist ? hist + (size_t)c * (Q - 1) : nullptr;
    for (int p = threadIdx.x; p < span; p += RS_THREADS) {
        const long idx = base + p;
        T v = 0;
        if (idx >= 0) {
            if (idx < n_in) v = row[idx];
        } else if (hrow && idx >= -(long)(Q - 1)) {
            v = hrow[(Q - 1) + idx];
        }
        xs[p] = v;
    }
    (void)span_max;
    __syncthreads();
    const long i = o0 + threadIdx.x;
    if (i >= n_out) return;
    const long long gi = i0 + i;
    const int newest = (int)((gi * M) / L - in0 - base);           // LDS index of x[(gi*M)/L]
    const typename rs_traits<T>::tap_t *grow = g + (size_t)(gi % L) * Q;
    acc_t acc = 0;
    if (sizeof(T) == sizeof(float)) {
        for (int k = 0; k < Q; k++) acc = (acc_t)__builtin_fmaf((float)xs[newest - k], (float)grow[k], (float)acc);
    } else {
        for (int k = 0; k < Q; k++) {                              // llz_resample.c:590-592, exact order
            const acc_t prod = (acc_t)xs[newest - k] * (acc_t)grow[k];
            acc = acc + prod;
        }
    }
    out[(size_t)c * out_pitch + i] = rs_finish(acc, gain, (T *)nullptr);
}

// history update for int16 (float uses k_fir_tail_f32): hist_new = last `keep` of concat(hist_old, in[0:n])
__global__ void __launch_bounds__(256)
k_tail_i16(const short *__restrict__ in, const short *__restrict__ hist_old, short *__restrict__ hist_new, long n,
           long in_pitch, int keep)
{
    const int c = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= keep) return;
    const long idx = n - keep + j;
    hist_new[(size_t)c * keep + j] = idx >= 0 ? in[(size_t)c * in_pitch + idx]
                                              : hist_old[(size_t)c * keep + (keep + idx)];
}

// llz_decimate inner loops (llz_resample.c:457-483): buf = [n history samples | num_in new samples];
// y_i = sum_{m<M} sum_{k<K} buf[i*M + m + M*k] * p[m][k], m outer, k inner, one rounded multiply and add per term
__global__ void __launch_bounds__(256)
k_decimate_i16(const short *__restrict__ buf, short *__restrict__ out, const double *__restrict__ p, int M, int K,
               int num_out, double gain)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= num_out) return;
    const short *x = buf + (size_t)i * M;
    double y = 0.0;
    for (int m = 0; m < M; m++)
        for (int k = 0; k < K; k++) {
            const double prod = (double)x[m + M * k] * p[(size_t)m * K + k];
            y = y + prod;
        }
    out[i] = rs_finish(y, gain, (short *)nullptr);
}

// llz_interp inner loops (llz_resample.c:515-536): x has K zero samples behind the frame (the reference reads
// past its input there); out[i*L + (L-1-m)] = clamp(gain * sum_k x[i+k] * p[m][k])
__global__ void __launch_bounds__(256)
k_interp_i16(const short *__restrict__ x, short *__restrict__ out, const double *__restrict__ p, int L, int K,
             int num_in, double gain)
{
#pragma clang fp contract(off)
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= num_in * L) return;
    const int i = t / L, m = t - i * L;
    double y = 0.0;
    for (int k = 0; k < K; k++) {
        const double prod = (double)x[i + k] * p[(size_t)m * K + k];
        y = y + prod;
    }
    out[(size_t)i * L + (L - 1 - m)] = rs_finish(y, gain, (short *)nullptr);
}

template <typename T>
int launch_resample(const T *in, T *out, const T *hist, const typename rs_traits<T>::tap_t *g, int channels,
                    long n_in, long n_out, long in_pitch, long out_pitch, int L, int M, int Q,
                    typename rs_traits<T>::acc_t gain, long long i0, long long in0, void *stream, const char *name)
{
    if (!in || !out || !g || channels <= 0 || channels > 65535 || n_in <= 0 || n_out <= 0 || L < 1 || M < 1 ||
        Q < 1 || in_pitch < n_in || out_pitch < n_out) {
        llzs_set_error("%s: bad arguments", name);
        return LLZ_ERR_ARG;
    }
    // newest-input positions of 256 consecutive outputs span at most ceil(255*M/L)+1 samples
    const int span_max = (int)((255L * M + L - 1) / L) + 1 + Q;
    const size_t lds = (size_t)span_max * sizeof(T);
    if (lds > 160 * 1024) {
        llzs_set_error("%s: L=%d M=%d Q=%d needs %zu B of LDS", name, L, M, Q, lds);
        return LLZ_ERR_RANGE;
    }
    if (lds > 64 * 1024)
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resample<T>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)((n_out + RS_THREADS - 1) / RS_THREADS), (unsigned)channels);
    hipLaunchKernelGGL(k_resample<T>, grid, dim3(RS_THREADS), lds, as_stream(stream), in, out, hist, g, n_in,
                       n_out, in_pitch, out_pitch, L, M, Q, gain, i0, in0, span_max);
    LLZ_LAUNCH_CHECK(name);
    return LLZ_OK;
}

} // namespace

extern "C" int llzs_resample_f32(const float *in, float *out, const float *hist, const float *g, int channels,
                                 long n_in, long n_out, long in_pitch, long out_pitch, int L, int M, int Q,
                                 float gain, long long i0, long long in0, void *stream)
{
    return launch_resample<float>(in, out, hist, g, channels, n_in, n_out, in_pitch, out_pitch, L, M, Q, gain, i0,
                                  in0, stream, "k_resample<float>");
}

extern "C" int llzs_resample_i16(const short *in, short *out, const short *hist, const double *g, int channels,
                                 long n_in, long n_out, long in_pitch, long out_pitch, int L, int M, int Q,
                                 double gain, long long i0, long long in0, void *stream)
{
    return launch_resample<short>(in, out, hist, g, channels, n_in, n_out, in_pitch, out_pitch, L, M, Q, gain, i0,
                                  in0, stream, "k_resample<short>");
}

extern "C" int llzs_tail_i16(const short *in, const short *hist_old, short *hist_new, int channels, long n,
                             long in_pitch, int keep, void *stream)
{
    if (keep <= 0) return LLZ_OK;
    if (!in || !hist_old || !hist_new || channels <= 0 || channels > 65535 || n <= 0) {
        llzs_set_error("tail_i16: bad arguments");
        return LLZ_ERR_ARG;
    }
    dim3 grid((unsigned)((keep + 255) / 256), (unsigned)channels);
    hipLaunchKernelGGL(k_tail_i16, grid, dim3(256), 0, as_stream(stream), in, hist_old, hist_new, n, in_pitch,
                       keep);
    LLZ_LAUNCH_CHECK("k_tail_i16");
    return LLZ_OK;
}

extern "C" int llzs_decimate_i16(const short *buf, short *out, const double *p, int M, int K, int n, int num_out,
                                 double gain, void *stream)
{
    (void)n;
    if (!buf || !out || !p || M < 1 || K < 1 || num_out < 1) {
        llzs_set_error("decimate_i16: bad arguments");
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_decimate_i16, dim3((unsigned)((num_out + 255) / 256)), dim3(256), 0, as_stream(stream),
                       buf, out, p, M, K, num_out, gain);
    LLZ_LAUNCH_CHECK("k_decimate_i16");
    return LLZ_OK;
}

extern "C" int llzs_interp_i16(const short *x, short *out, const double *p, int L, int K, int num_in, double gain,
                               void *stream)
{
    if (!x || !out || !p || L < 1 || K < 1 || num_in < 1) {
        llzs_set_error("interp_i16: bad arguments");
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_interp_i16, dim3((unsigned)((num_in * L + 255) / 256)), dim3(256), 0, as_stream(stream),
                       x, out, p, L, K, num_in, gain);
    LLZ_LAUNCH_CHECK("k_interp_i16");
    return LLZ_OK;
}
