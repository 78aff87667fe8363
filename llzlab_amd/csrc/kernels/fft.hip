// fft.hip -- K4/K5 standalone: batched radix-2 complex FFT in LDS for gfx950, three arithmetic flavours that share
// the reference's exact dataflow (reference libllzfilter/llz_fft.c:61-198, llz_fft_fixed.c:61-218):
//
//   forward : DIF butterflies, half-span = N/2 ... 1, twiddle index q * (N / span), w = cos - j sin, no scaling,
//             then the bit-reversal gather to natural order;
//   inverse : bit-reversal gather (float: each element divided by N there), DIT butterflies half-span 1 ... N/2,
//             w = cos + j sin (fixed point: arithmetic >> log2 N once at the very end).
//
//   float   tolerance path (batched float32, the overlap-save building block)
//   double  the reference's own arithmetic, rounded multiply / add in its expression order, no contraction:
//           bit-identical to llz_fft / llz_ifft for the same host-built twiddle table
//   int32   Q15 twiddles, (int64 a * b) >> 15 per product, wrapping adds: bit-identical to llz_fft_fixed
//
// One workgroup per transform; the whole transform lives in LDS (N <= 4096: 32 KB float / 64 KB double);
// butterflies of a stage are independent, so only a barrier separates stages. Twiddle tables come from the host
// (never recomputed on the device: SURVEY.md H4/H5).
#include "common.hpp"

namespace {

constexpr int FFT_THREADS = 256;

template <typename T>
struct cpx {
    T re, im;
};

struct arith_f32 {
    typedef float data_t;
    typedef float tw_t;
    static __device__ __forceinline__ float add(float a, float b) { return a + b; }
    static __device__ __forceinline__ float sub(float a, float b) { return a - b; }
    // (dr*wr - di*wi, dr*wi + di*wr)
    static __device__ __forceinline__ void rot(float dr, float di, float wr, float wi, float &yr, float &yi)
    {
        yr = __builtin_fmaf(dr, wr, -(di * wi));
        yi = __builtin_fmaf(dr, wi, di * wr);
    }
    static __device__ __forceinline__ float neg(float w) { return -w; }
    static __device__ __forceinline__ float scale_in(float v, int n, int) { return v / (float)n; }
    static __device__ __forceinline__ float scale_out(float v, int) { return v; }
};

struct arith_f64 {
    typedef double data_t;
    typedef double tw_t;
    static __device__ __forceinline__ double add(double a, double b)
    {
#pragma clang fp contract(off)
        return a + b;
    }
    static __device__ __forceinline__ double sub(double a, double b)
    {
#pragma clang fp contract(off)
        return a - b;
    }
    static __device__ __forceinline__ void rot(double dr, double di, double wr, double wi, double &yr, double &yi)
    {
#pragma clang fp contract(off)
        const double p0 = dr * wr, p1 = di * wi, p2 = dr * wi, p3 = di * wr;   // llz_fft.c:81-82 / :122-123
        yr = p0 - p1;
        yi = p2 + p3;
    }
    static __device__ __forceinline__ double neg(double w) { return -w; }
    static __device__ __forceinline__ double scale_in(double v, int n, int)
    {
#pragma clang fp contract(off)
        return v / (double)n;                                                    // llz_fft.c:193-194: true division
    }
    static __device__ __forceinline__ double scale_out(double v, int) { return v; }
};

struct arith_q15 {
    typedef int data_t;
    typedef short tw_t;
    static __device__ __forceinline__ int add(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
    static __device__ __forceinline__ int sub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
    static __device__ __forceinline__ int mul15(int a, short b)
    {
        return (int)(((long long)a * (long long)b) >> 15);                       // llz_fft_fixed.h:67
    }
    static __device__ __forceinline__ void rot(int dr, int di, short wr, short wi, int &yr, int &yi)
    {
        yr = sub(mul15(dr, wr), mul15(di, wi));                                  // llz_fft_fixed.c:86-87
        yi = add(mul15(dr, wi), mul15(di, wr));
    }
    static __device__ __forceinline__ short neg(short w) { return (short)(-w); }
    static __device__ __forceinline__ int scale_in(int v, int, int) { return v; }
    static __device__ __forceinline__ int scale_out(int v, int log2n) { return v >> log2n; }   // :212-215
};

template <typename A, bool INVERSE>
__global__ void __launch_bounds__(FFT_THREADS)
k_fft_radix2(typename A::data_t *__restrict__ data, int size, int log2n,
             const typename A::tw_t *__restrict__ cs /* size cos, then size sin */)
{
    typedef typename A::data_t T;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cpx<T> *s = reinterpret_cast<cpx<T> *>(smem_raw);
    cpx<T> *g = reinterpret_cast<cpx<T> *>(data) + (size_t)blockIdx.x * size;
    const int tid = threadIdx.x;
    const int half_n = size >> 1;

    if (!INVERSE) {
        for (int i = tid; i < size; i += FFT_THREADS) s[i] = g[i];
        __syncthreads();
        // span = distance between partners; twiddle step doubles as the span halves
        for (int hs = half_n, tstep = 1; hs >= 1; hs >>= 1, tstep <<= 1) {
            for (int b = tid; b < half_n; b += FFT_THREADS) {
                const int q = b & (hs - 1);
                const int lo = ((b - q) << 1) + q, hi = lo + hs;
                const typename A::tw_t wr = cs[q * tstep], wi = A::neg(cs[size + q * tstep]);
                const cpx<T> u = s[lo], v = s[hi];
                cpx<T> x, y;
                x.re = A::add(u.re, v.re); x.im = A::add(u.im, v.im);
                A::rot(A::sub(u.re, v.re), A::sub(u.im, v.im), wr, wi, y.re, y.im);
                s[lo] = x; s[hi] = y;
            }
            __syncthreads();
        }
        for (int i = tid; i < size; i += FFT_THREADS)
            g[i] = s[__brev((unsigned)i) >> (32 - log2n)];
    } else {
        for (int i = tid; i < size; i += FFT_THREADS) {
            cpx<T> v = g[__brev((unsigned)i) >> (32 - log2n)];
            v.re = A::scale_in(v.re, size, log2n);
            v.im = A::scale_in(v.im, size, log2n);
            s[i] = v;
        }
        __syncthreads();
        for (int hs = 1, tstep = half_n; hs <= half_n; hs <<= 1, tstep >>= 1) {
            for (int b = tid; b < half_n; b += FFT_THREADS) {
                const int q = b & (hs - 1);
                const int lo = ((b - q) << 1) + q, hi = lo + hs;
                const typename A::tw_t wr = cs[q * tstep], wi = cs[size + q * tstep];
                const cpx<T> u = s[lo], v = s[hi];
                T dr, di;
                A::rot(v.re, v.im, wr, wi, dr, di);
                cpx<T> x, y;
                x.re = A::add(u.re, dr); x.im = A::add(u.im, di);
                y.re = A::sub(u.re, dr); y.im = A::sub(u.im, di);
                s[lo] = x; s[hi] = y;
            }
            __syncthreads();
        }
        for (int i = tid; i < size; i += FFT_THREADS) {
            cpx<T> v = s[i];
            v.re = A::scale_out(v.re, log2n);
            v.im = A::scale_out(v.im, log2n);
            g[i] = v;
        }
    }
}

template <typename A>
int launch_fft(typename A::data_t *data, int count, int size, const typename A::tw_t *cs, int inverse,
               void *stream, const char *name)
{
    int log2n = 0;
    while ((1 << log2n) < size) log2n++;
    if (!data || !cs || count < 1 || size < 2 || size > 4096 || (1 << log2n) != size) {
        llzs_set_error("%s: size %d must be a power of two in 2..4096 (count %d)", name, size, count);
        return LLZ_ERR_ARG;
    }
    const size_t lds = (size_t)size * 2 * sizeof(typename A::data_t);
    if (lds >= 64 * 1024) {
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft_radix2<A, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft_radix2<A, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (inverse)
        hipLaunchKernelGGL((k_fft_radix2<A, true>), dim3((unsigned)count), dim3(FFT_THREADS), lds,
                           as_stream(stream), data, size, log2n, cs);
    else
        hipLaunchKernelGGL((k_fft_radix2<A, false>), dim3((unsigned)count), dim3(FFT_THREADS), lds,
                           as_stream(stream), data, size, log2n, cs);
    LLZ_LAUNCH_CHECK(name);
    return LLZ_OK;
}

} // namespace

extern "C" int llzs_fft_f32(float *data, int count, int size, const float *cs, int inverse, void *stream)
{
    return launch_fft<arith_f32>(data, count, size, cs, inverse, stream, "k_fft_radix2<f32>");
}

extern "C" int llzs_fft_f64(double *data, int size, const double *cs, int inverse, void *stream)
{
    return launch_fft<arith_f64>(data, 1, size, cs, inverse, stream, "k_fft_radix2<f64>");
}

extern "C" int llzs_fft_fixed(int *data, int count, int size, const short *cs, int inverse, void *stream)
{
    return launch_fft<arith_q15>(data, count, size, cs, inverse, stream, "k_fft_radix2<q15>");
}
