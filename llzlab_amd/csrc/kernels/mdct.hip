// mdct.hip -- MDCT kernels for gfx950 (reference libllzfilter/llz_mdct.c:185-353).
//
//  k_matvec_exact_f64   y[r] = sum_c x[c] * A[r][c] in ascending c with a rounded multiply and a rounded add per term:
//                       the defining sums of the reference's MDCT_ORIGIN type (mdct0 / imdct0, llz_mdct.c:185-222) in the
//                       reference's operation order, one lane per output (bit-identical; exists for parity, not speed).
//  k_mdct4_f32          the N/4-point-FFT algorithm (mdct2 / imdct2, llz_mdct.c:266-353) for many frames at once in
//                       float32: rotate + pre-twiddle into LDS, the shared float32 FFT passes, post-twiddle, and the
//                       output permutation staged through LDS so that HBM sees contiguous rows on both sides.
#include "fft_core.hpp"

namespace {

__global__ void __launch_bounds__(256)
k_matvec_exact_f64(const double *__restrict__ A, const double *__restrict__ x, double *__restrict__ y, int rows,
                   int cols)
{
#pragma clang fp contract(off)
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const double *row = A + (size_t)r * cols;
    double acc = 0;
    for (int c = 0; c < cols; c++) {
        const double prod = x[c] * row[c];
        acc = acc + prod;
    }
    y[r] = acc;
}

// y[r] = sum_c (int)(((int64)x[c] * A[r][c]) >> 15) with wrapping int32 adds: the defining sums of MDCT_FIXED_ORIGIN
// (llz_mdct_fixed.c:116-152).  Integer adds are associative modulo 2^32, so any order is the reference's result.
__global__ void __launch_bounds__(256)
k_matvec_q15(const short *__restrict__ A, const int *__restrict__ x, int *__restrict__ y, int rows, int cols)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const short *row = A + (size_t)r * cols;
    unsigned acc = 0;
    for (int c = 0; c < cols; c++) acc += (unsigned)(int)(((long long)x[c] * (long long)row[c]) >> 15);
    y[r] = (int)acc;
}

// One workgroup transforms tpw frames of length N (N4 = N/4 complex points each).
//   forward: in = x [count][N], out = X [count][N/2];   inverse: in = X [count][N/2], out = x [count][N]
// tc/ts: cos and sin of -2 pi (k + 1/8) / N, k < N/4 (llz_mdct.c:459-462); cs: FFT table of size N/4.
template <bool INVERSE>
__global__ void __launch_bounds__(FFT_THREADS)
k_mdct4_f32(const float *__restrict__ in, float *__restrict__ out, int count, int N, int log2n4,
            const float *__restrict__ tc, const float *__restrict__ ts, const float *__restrict__ cs, int tpw,
            unsigned groups, float sqrt_cof)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int N2 = N >> 1, N4 = N >> 2;
    const int tstride = fft_phys(N4) + 1;
    cpx<float> *s = reinterpret_cast<cpx<float> *>(smem_raw);
    cpx<float> *tw = s + (size_t)tpw * tstride;
    float *buf = reinterpret_cast<float *>(tw + tw_entries(N4));            // [tpw][N] staging of rows
    const int tid = threadIdx.x;
    const int tr0 = blockIdx.x * tpw;
    const int ntr = min(tpw, count - tr0);
    const int in_len = INVERSE ? N2 : N, out_len = INVERSE ? N : N2;
    fft_load_twiddles(tw, cs, N4, tid);
    for (int e = tid; e < ntr * in_len; e += FFT_THREADS) {                 // contiguous rows in
        const int tr = e / in_len, i = e - tr * in_len;
        buf[tr * N + i] = in[(size_t)(tr0 + tr) * in_len + i];
    }
    __syncthreads();
    // pre-twiddle: z[k] = 0.5 * (re + j im) * (c + j s)
    for (int e = tid; e < ntr * N4; e += FFT_THREADS) {
        const int tr = e >> log2n4, k = e & (N4 - 1);
        const float *x = buf + tr * N;
        float re, im;
        if (!INVERSE) {
            // rot[i] = -x[i + 3N/4] (i < N/4), x[i - N/4] otherwise (llz_mdct.c:279-283)
            auto rot = [&](int i) { return i < N4 ? -x[i + 3 * N4] : x[i - N4]; };
            re = rot(2 * k) - rot(N - 1 - 2 * k);
            im = rot(N2 - 1 - 2 * k) - rot(N2 + 2 * k);
        } else {
            re = x[2 * k];                                                   // llz_mdct.c:322-324
            im = x[N2 - 1 - 2 * k];
        }
        const float c = tc[k], sn = ts[k];
        cpx<float> z;
        z.re = 0.5f * (re * c - im * sn);
        z.im = 0.5f * (re * sn + im * c);
        s[tr * tstride + fft_phys(k)] = z;
    }
    __syncthreads();
    fft_run_f32<false>(s, ntr, N4, log2n4, tstride, tw, groups, tid);       // both directions use the FORWARD transform
    // post-twiddle from bit-reversed positions into the staging rows, in output order
    for (int e = tid; e < ntr * N4; e += FFT_THREADS) {
        const int tr = e >> log2n4, k = e & (N4 - 1);
        const cpx<float> v = s[tr * tstride + fft_phys((int)(__brev((unsigned)k) >> (32 - log2n4)))];
        const float c = tc[k], sn = ts[k];
        float *y = buf + tr * N;
        if (!INVERSE) {
            y[2 * k] = 2.f * (v.re * c - v.im * sn);                        // llz_mdct.c:296-301
            y[N2 - 1 - 2 * k] = -2.f * (v.re * sn + v.im * c);
        } else {
            // rot[2k] = re', rot[N/2 + 2k] = im', rot[odd i] = -rot[N-1-i]; x[i] = rot[N/4 + i] * cof (i < 3N/4),
            // -rot[i - 3N/4] * cof otherwise (llz_mdct.c:331-352): scatter each value to the (up to two) x it feeds
            const float re = 8.f * sqrt_cof * (v.re * c - v.im * sn);
            const float im = 8.f * sqrt_cof * (v.re * sn + v.im * c);
            auto put = [&](int ri, float val) {                             // rot[ri] = val -> x
                if (ri >= N4) y[ri - N4] = val * sqrt_cof;
                else y[ri + 3 * N4] = -val * sqrt_cof;
            };
            put(2 * k, re);
            put(N - 1 - 2 * k, -re);                                         // odd index N-1-2k mirrors rot[2k]
            put(N2 + 2 * k, im);
            put(N2 - 1 - 2 * k, -im);                                        // odd index N/2-1-2k mirrors rot[N/2+2k]
        }
    }
    __syncthreads();
    for (int e = tid; e < ntr * out_len; e += FFT_THREADS) {                // contiguous rows out
        const int tr = e / out_len, i = e - tr * out_len;
        out[(size_t)(tr0 + tr) * out_len + i] = buf[tr * N + i];
    }
}

} // namespace

extern "C" int llzs_matvec_exact_f64(const double *A, const double *x, double *y, int rows, int cols, void *stream)
{
    if (!A || !x || !y || rows < 1 || cols < 1) {
        llzs_set_error("matvec_exact_f64: bad arguments");
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_matvec_exact_f64, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream), A, x, y,
                       rows, cols);
    LLZ_LAUNCH_CHECK("k_matvec_exact_f64");
    return LLZ_OK;
}

extern "C" int llzs_matvec_q15(const short *A, const int *x, int *y, int rows, int cols, void *stream)
{
    if (!A || !x || !y || rows < 1 || cols < 1) {
        llzs_set_error("matvec_q15: bad arguments");
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_matvec_q15, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream), A, x, y, rows,
                       cols);
    LLZ_LAUNCH_CHECK("k_matvec_q15");
    return LLZ_OK;
}

// in/out: device, contiguous rows; N a power of two in 32..8192; tc, ts: N/4 floats; cs: 2 * (N/4) floats
extern "C" int llzs_mdct4_f32(const float *in, float *out, int count, int N, const float *tc, const float *ts,
                              const float *cs, int inverse, void *stream)
{
    int log2n = 0;
    while ((1 << log2n) < N) log2n++;
    if (!in || !out || !tc || !ts || !cs || count < 1 || N < 32 || N > 8192 || (1 << log2n) != N) {
        llzs_set_error("mdct4_f32: bad arguments (N=%d must be a power of two in 32..8192, count=%d)", N, count);
        return LLZ_ERR_ARG;
    }
    if (llzs_tune(LLZS_TUNE_FFT_GENERIC) < 1) {                              // the six sizes with a register-transform kernel
        const int rc = llzs_mdct4_reg_f32(in, out, count, N, tc, ts, cs, inverse, stream);
        if (rc != LLZ_ERR_RANGE) return rc;
    }
    const int N4 = N >> 2, log2n4 = log2n - 2;
    int tpw = 2048 / N4;
    if (tpw > count) tpw = count;
    const int tstride = N4 + (N4 >> 5) + 1;
    const size_t lds = (size_t)tpw * tstride * 2 * sizeof(float) + (size_t)tw_entries(N4) * 2 * sizeof(float) +
                       (size_t)tpw * N * sizeof(float);
    const unsigned blocks = (unsigned)((count + tpw - 1) / tpw);
    const float sqrt_cof = (float)(1.0 / sqrt((double)N));
    if (inverse) {
        if (lds >= 64 * 1024)
            LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mdct4_f32<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mdct4_f32<true>, dim3(blocks), dim3(FFT_THREADS), lds, as_stream(stream), in, out, count, N,
                           log2n4, tc, ts, cs, tpw, fft_groups(log2n4), sqrt_cof);
    } else {
        if (lds >= 64 * 1024)
            LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mdct4_f32<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mdct4_f32<false>, dim3(blocks), dim3(FFT_THREADS), lds, as_stream(stream), in, out, count, N,
                           log2n4, tc, ts, cs, tpw, fft_groups(log2n4), sqrt_cof);
    }
    LLZ_LAUNCH_CHECK("k_mdct4_f32");
    return LLZ_OK;
}
