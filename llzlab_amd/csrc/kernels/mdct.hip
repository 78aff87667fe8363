// mdct.hip -- MDCT kernels for gfx950 (reference libllzfilter/llz_mdct.c:185-353).
//
//  k_matvec_exact_f64   y[r] = sum_c x[c] * A[r][c] in ascending c with a rounded multiply and a rounded add per term:
//                       the defining sums of the reference's MDCT_ORIGIN type (mdct0 / imdct0, llz_mdct.c:185-222) in the
//                       reference's operation order, one lane per output (bit-identical; exists for parity, not speed).
//  k_mdct_rot_*_f64     the twiddle steps of the reference's two FFT forms (mdct1 / imdct1 around an N-point transform,
//                       mdct2 / imdct2 around an N/4-point one, llz_mdct.c:225-353) in double with the reference's
//                       rounding order (every product and every sum rounded, no contraction), one lane per output:
//                       together with the exact-order transform (fft.hip) the whole single-channel MDCT runs on the
//                       device and stays bit-identical.
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

// ---- exact-order twiddle steps of the FFT forms (double, one frame) ----------------------------------------------------
// Tables are (cos, sin) pairs.  "N-point form" (llz_mdct.c:225-264): pre = modulate the frame into N complex points, post =
// rotate each bin and keep the real part.  "quarter form" (llz_mdct.c:266-353): pre = fold the frame into N/4 complex
// points and rotate, post = rotate and scatter.
__global__ void __launch_bounds__(256)
k_mdct_rot_full_pre_f64(const double *__restrict__ in, double *__restrict__ z, const double *__restrict__ cs2, int N,
                        int inverse)
{
#pragma clang fp contract(off)
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    // forward: point k = x[k] * (c, s)[k];  inverse: the N/2 coefficients extended by odd symmetry, X[k] for k < N/2 and
    // -X[N-1-k] above
    double v;
    if (!inverse) v = in[k];
    else v = k < (N >> 1) ? in[k] : -in[N - 1 - k];
    z[2 * k] = v * cs2[2 * k];
    z[2 * k + 1] = v * cs2[2 * k + 1];
}

__global__ void __launch_bounds__(256)
k_mdct_rot_full_post_f64(const double *__restrict__ z, double *__restrict__ out, const double *__restrict__ cs2, int N,
                         int inverse)
{
#pragma clang fp contract(off)
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= (inverse ? N : (N >> 1))) return;
    const double a = z[2 * k] * cs2[2 * k];
    const double b = z[2 * k + 1] * cs2[2 * k + 1];
    const double d = a - b;
    out[k] = inverse ? 2 * d : d;
}

__device__ __forceinline__ void mdct_quarter_rotate(double re, double im, double c, double s, double scale, double *zr,
                                                    double *zi)
{
#pragma clang fp contract(off)
    const double p = re * c, q = im * s, u = re * s, w = im * c;
    const double dr = p - q, di = u + w;
    *zr = scale * dr;
    *zi = scale * di;
}

__global__ void __launch_bounds__(256)
k_mdct_rot_quarter_pre_f64(const double *__restrict__ in, double *__restrict__ z, const double *__restrict__ cs2, int N,
                           int inverse)
{
#pragma clang fp contract(off)
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int N2 = N >> 1, N4 = N >> 2;
    if (k >= N4) return;
    double re, im;
    if (!inverse) {
        // the frame rotated by N/4 with its last quarter negated in front, folded onto N/4 points
        auto rotated = [&](int i) { return i < N4 ? -in[i + 3 * N4] : in[i - N4]; };
        re = rotated(2 * k) - rotated(N - 1 - 2 * k);
        im = rotated(N2 - 1 - 2 * k) - rotated(N2 + 2 * k);
    } else {
        re = in[2 * k];
        im = in[N2 - 1 - 2 * k];
    }
    mdct_quarter_rotate(re, im, cs2[2 * k], cs2[2 * k + 1], 0.5, &z[2 * k], &z[2 * k + 1]);
}

// forward: out = X [N/2]; inverse: out = x [N], one lane per time sample
__global__ void __launch_bounds__(256)
k_mdct_rot_quarter_post_f64(const double *__restrict__ z, double *__restrict__ out, const double *__restrict__ cs2, int N,
                            int inverse, double cof)
{
#pragma clang fp contract(off)
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int N2 = N >> 1, N4 = N >> 2;
    if (!inverse) {
        if (t >= N4) return;
        double zr, zi;
        mdct_quarter_rotate(z[2 * t], z[2 * t + 1], cs2[2 * t], cs2[2 * t + 1], 1.0, &zr, &zi);   // x 1.0 is exact
        out[2 * t] = 2 * zr;
        out[N2 - 1 - 2 * t] = -2 * zi;
        return;
    }
    if (t >= N) return;
    // the rotated sequence r: r[2m] = Re w[m], r[N/2 + 2m] = Im w[m] (w = 8 cof times the rotated bins), odd entries by
    // r[i] = -r[N-1-i]; the frame is r shifted back by N/4 with the wrapped quarter negated, times cof
    const double scale = 8 * cof;
    auto r_even = [&](int i) {                                  // i even
        const int m = i < N2 ? i >> 1 : (i - N2) >> 1;
        double zr, zi;
        mdct_quarter_rotate(z[2 * m], z[2 * m + 1], cs2[2 * m], cs2[2 * m + 1], scale, &zr, &zi);
        return i < N2 ? zr : zi;
    };
    auto r_at = [&](int i) { return (i & 1) ? -r_even(N - 1 - i) : r_even(i); };
    out[t] = t < 3 * N4 ? r_at(N4 + t) * cof : -r_at(t - 3 * N4) * cof;
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

extern "C" int llzs_mdct_rot_f64(int quarter, int post, const double *in, double *out, const double *cs2, int N, int inverse,
                                 double cof, void *stream)
{
    if (!in || !out || !cs2 || N < 4) {
        llzs_set_error("mdct_rot_f64: bad arguments");
        return LLZ_ERR_ARG;
    }
    const dim3 grid((unsigned)((N + 255) / 256)), block(256);
    if (!quarter && !post)
        hipLaunchKernelGGL(k_mdct_rot_full_pre_f64, grid, block, 0, as_stream(stream), in, out, cs2, N, inverse);
    else if (!quarter)
        hipLaunchKernelGGL(k_mdct_rot_full_post_f64, grid, block, 0, as_stream(stream), in, out, cs2, N, inverse);
    else if (!post)
        hipLaunchKernelGGL(k_mdct_rot_quarter_pre_f64, grid, block, 0, as_stream(stream), in, out, cs2, N, inverse);
    else
        hipLaunchKernelGGL(k_mdct_rot_quarter_post_f64, grid, block, 0, as_stream(stream), in, out, cs2, N, inverse, cof);
    LLZ_LAUNCH_CHECK("k_mdct_rot_f64");
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
