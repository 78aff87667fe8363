// mdct_q15.hip -- the fixed-point MDCT's own steps on the device (fixed half; reference
// libllzfilter/llz_mdct_fixed.c:116-283), for any number of frames per launch (grid.y = frame).
//
// Data are int32, tables Q15.  Every product is the reference's LLZ_FIXMUL_32X15 (llz_fft_fixed.h:67):
// (int)(((int64)a * (int64)b) >> 15), floored separately; sums, differences and negations wrap modulo 2^32 as the
// reference's int arithmetic does on this ABI (written on unsigned operands here, so the wrap is defined).  Integer
// arithmetic has no rounding order: one lane per output computing the output's own expression IS the reference's result,
// whatever the reference's loop structure was.  The transform between the two steps is the bit-exact Q15 FFT of fft.hip.
//
//  k_mdctq_sums        MDCT_FIXED_ORIGIN: the defining sums (a Q15 cosine matrix times the frame), optional (4 v) / N
//  k_mdctq_modulate    MDCT_FIXED_FFT, in front of the N-point transform: sample k -> complex point k
//  k_mdctq_demodulate  MDCT_FIXED_FFT, behind it: rotate a bin, keep the real part
//  k_mdctq_fold        MDCT_FIXED_FFT4, in front of the N/4-point transform: fold the frame to N/4 points, rotate, halve
//  k_mdctq_unfold      MDCT_FIXED_FFT4, behind it: rotate and scatter (forward) / rebuild the time frame (inverse)
//  k_mdct4_q15         MDCT_FIXED_FFT4 in ONE launch: fold, the N/4-point Q15 transform and unfold in LDS (rows in, rows out)
//  k_mdct1_q15         MDCT_FIXED_FFT in ONE launch: modulate, the N-point Q15 transform and demodulate in LDS
#include "common.hpp"
#include "fft_core.hpp"

namespace {

__device__ __forceinline__ int q15(int a, int b) { return arith_q15::mul15(a, (short)b); }     // (b: a Q15 table value)
__device__ __forceinline__ int wrap_add(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int wrap_sub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int wrap_neg(int a) { return (int)(0u - (unsigned)a); }
__device__ __forceinline__ int wrap_scale(int a, int by) { return (int)((unsigned)a * (unsigned)by); }

// y[f][r] = sum_c q15(x[f][c], A[r][c]); with quarter_over_n the reference's "(sum * 4) / N" (wrapping product, C division)
__global__ void __launch_bounds__(256)
k_mdctq_sums(const short *__restrict__ A, const int *__restrict__ x, int *__restrict__ y, int rows, int cols,
             int quarter_over_n)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const short *row = A + (size_t)r * cols;
    const int *frame = x + (size_t)blockIdx.y * cols;
    unsigned acc = 0;
    for (int c = 0; c < cols; c++) acc += (unsigned)q15(frame[c], row[c]);
    int v = (int)acc;
    if (quarter_over_n) v = wrap_scale(v, 4) / quarter_over_n;
    y[(size_t)blockIdx.y * rows + r] = v;
}

// forward: frame [N] -> points [N] complex; inverse: coefficients [N/2], continued with odd symmetry (X[k], then -X[N-1-k])
__global__ void __launch_bounds__(256)
k_mdctq_modulate(const int *__restrict__ in, int *__restrict__ z, const short2 *__restrict__ tw, int N, int inverse)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= N) return;
    const int *src = in + (size_t)blockIdx.y * (inverse ? N >> 1 : N);
    int v;
    if (!inverse) v = src[k];
    else v = k < (N >> 1) ? src[k] : wrap_neg(src[N - 1 - k]);
    const short2 w = tw[k];
    int2 p;
    p.x = q15(v, w.x);
    p.y = q15(v, w.y);
    reinterpret_cast<int2 *>(z)[(size_t)blockIdx.y * N + k] = p;
}

// forward: N/2 coefficients; inverse: N samples, doubled
__global__ void __launch_bounds__(256)
k_mdctq_demodulate(const int *__restrict__ z, int *__restrict__ out, const short2 *__restrict__ tw, int N, int inverse)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int n_out = inverse ? N : N >> 1;
    if (k >= n_out) return;
    const int2 p = reinterpret_cast<const int2 *>(z)[(size_t)blockIdx.y * N + k];
    const short2 w = tw[k];
    const int d = wrap_sub(q15(p.x, w.x), q15(p.y, w.y));
    out[(size_t)blockIdx.y * n_out + k] = inverse ? (int)((unsigned)d << 1) : d;
}

// (re + j im)(c + j s) with four separately floored products
__device__ __forceinline__ void q15_rotate(int re, int im, short2 w, int *zr, int *zi)
{
    *zr = wrap_sub(q15(re, w.x), q15(im, w.y));
    *zi = wrap_add(q15(re, w.y), q15(im, w.x));
}

__global__ void __launch_bounds__(256)
k_mdctq_fold(const int *__restrict__ in, int *__restrict__ z, const short2 *__restrict__ tw, int N, int inverse)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int N2 = N >> 1, N4 = N >> 2;
    if (k >= N4) return;
    const int *src = in + (size_t)blockIdx.y * (inverse ? N2 : N);
    int re, im;
    if (!inverse) {
        // the frame turned by a quarter, the quarter that wraps to the front negated
        auto turned = [&](int i) { return i < N4 ? wrap_neg(src[i + 3 * N4]) : src[i - N4]; };
        re = wrap_sub(turned(2 * k), turned(N - 1 - 2 * k));
        im = wrap_sub(turned(N2 - 1 - 2 * k), turned(N2 + 2 * k));
    } else {
        re = src[2 * k];
        im = src[N2 - 1 - 2 * k];
    }
    int zr, zi;
    q15_rotate(re, im, tw[k], &zr, &zi);
    reinterpret_cast<int2 *>(z)[(size_t)blockIdx.y * N4 + k] = make_int2(zr >> 1, zi >> 1);
}

__global__ void __launch_bounds__(256)
k_mdctq_unfold(const int *__restrict__ z, int *__restrict__ out, const short2 *__restrict__ tw, int N, int inverse,
               int cof)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int N2 = N >> 1, N4 = N >> 2;
    const int2 *pts = reinterpret_cast<const int2 *>(z) + (size_t)blockIdx.y * N4;
    if (!inverse) {
        if (t >= N4) return;
        int zr, zi;
        q15_rotate(pts[t].x, pts[t].y, tw[t], &zr, &zi);
        int *X = out + (size_t)blockIdx.y * N2;
        X[2 * t] = wrap_scale(zr, 2);
        X[N2 - 1 - 2 * t] = wrap_scale(zi, -2);
        return;
    }
    if (t >= N) return;
    // the turned sequence r: r[2m] = 8 q15(Re v[m], cof), r[N/2 + 2m] = 8 q15(Im v[m], cof) with v = the rotated bins; odd
    // entries mirror the even ones with the sign flipped; the frame is r turned back by a quarter (wrapped quarter negated),
    // times cof
    auto r_even = [&](int i) {
        const int m = i < N2 ? i >> 1 : (i - N2) >> 1;
        int zr, zi;
        q15_rotate(pts[m].x, pts[m].y, tw[m], &zr, &zi);
        return wrap_scale(q15(i < N2 ? zr : zi, cof), 8);
    };
    auto r_at = [&](int i) { return (i & 1) ? wrap_neg(r_even(N - 1 - i)) : r_even(i); };
    out[(size_t)blockIdx.y * N + t] = t < 3 * N4 ? q15(r_at(N4 + t), cof) : q15(wrap_neg(r_at(t - 3 * N4)), cof);
}

// MDCT_FIXED_FFT4 whole (llz_mdct_fixed.c:200-283): a workgroup takes tpw frames, brings their rows into LDS with full-line
// reads, folds and rotates them into the transform's LDS image (k_mdctq_fold's expressions), runs the radix-2 passes of the
// bit-exact Q15 transform (fft_core.hpp: every butterfly is the reference's butterfly), rotates the bins and scatters them
// into output rows in LDS (k_mdctq_unfold's expressions, one rotation per bin instead of one per output), and writes the rows
// with full-line stores.  HBM sees the frame once in and the result once out: 6 B per sample against 14 with three launches.
//   forward: in = x [count][N], out = X [count][N/2];   inverse: in = X [count][N/2], out = x [count][N]
// pre / post: the two rotation tables of this direction (N/4 short2 each); cs: the transform's table (N/4 cos, N/4 sin).
template <bool INVERSE>
__global__ void __launch_bounds__(FFT_THREADS)
k_mdct4_q15(const int *__restrict__ in, int *__restrict__ out, int count, int N, int log2n4,
            const short2 *__restrict__ pre, const short2 *__restrict__ post, const short *__restrict__ cs, int tpw,
            unsigned groups, int cof)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int N2 = N >> 1, N4 = N >> 2;
    const int tstride = fft_phys(N4) + 1;
    cpx<int> *s = reinterpret_cast<cpx<int> *>(smem_raw);
    cpx<short> *tw = reinterpret_cast<cpx<short> *>(s + (size_t)tpw * tstride);
    int *buf = reinterpret_cast<int *>(tw + tw_entries(N4) + (tw_entries(N4) & 1));      // [tpw][N] rows
    const int tid = threadIdx.x;
    const int tr0 = blockIdx.x * tpw;
    const int ntr = min(tpw, count - tr0);
    const int in_len = INVERSE ? N2 : N, out_len = INVERSE ? N : N2;
    fft_load_twiddles(tw, cs, N4, tid);
    {
        const int *src = in + (size_t)tr0 * in_len;                       // the workgroup's rows are one contiguous range
        const int total = ntr * in_len;
        if (in_len == N) {
            for (int e = tid; e < total; e += FFT_THREADS) buf[e] = src[e];
        } else {
            for (int e = tid; e < total; e += FFT_THREADS) buf[((e >> (log2n4 + 1)) << (log2n4 + 2)) + (e & (N2 - 1))] = src[e];
        }
    }
    __syncthreads();
    for (int e = tid; e < ntr * N4; e += FFT_THREADS) {
        const int tr = e >> log2n4, k = e & (N4 - 1);
        const int *x = buf + tr * N;
        int re, im;
        if (!INVERSE) {
            auto turned = [&](int i) { return i < N4 ? wrap_neg(x[i + 3 * N4]) : x[i - N4]; };
            re = wrap_sub(turned(2 * k), turned(N - 1 - 2 * k));
            im = wrap_sub(turned(N2 - 1 - 2 * k), turned(N2 + 2 * k));
        } else {
            re = x[2 * k];
            im = x[N2 - 1 - 2 * k];
        }
        int zr, zi;
        q15_rotate(re, im, pre[k], &zr, &zi);
        cpx<int> z;
        z.re = zr >> 1;
        z.im = zi >> 1;
        s[tr * tstride + fft_phys(k)] = z;
    }
    __syncthreads();
    {
        int done = 0;                                                     // (both directions run the FORWARD transform)
#pragma unroll 1
        for (int p = 0; p < 4; p++) {
            const int G = (groups >> (4 * p)) & 15;
            if (G == 0) break;
            const int log2step = log2n4 - done - G;
            switch (G) {
            case 1: fft_pass<arith_q15, 1, false>(s, ntr, N4, log2n4, log2step, tstride, tw, tid); break;
            case 2: fft_pass<arith_q15, 2, false>(s, ntr, N4, log2n4, log2step, tstride, tw, tid); break;
            case 3: fft_pass<arith_q15, 3, false>(s, ntr, N4, log2n4, log2step, tstride, tw, tid); break;
            default: fft_pass<arith_q15, 4, false>(s, ntr, N4, log2n4, log2step, tstride, tw, tid); break;
            }
            done += G;
        }
    }
    for (int e = tid; e < ntr * N4; e += FFT_THREADS) {
        const int tr = e >> log2n4, m = e & (N4 - 1);
        const cpx<int> v = s[tr * tstride + fft_phys((int)(__brev((unsigned)m) >> (32 - log2n4)))];   // bin m
        int zr, zi;
        q15_rotate(v.re, v.im, post[m], &zr, &zi);
        int *y = buf + tr * N;
        if (!INVERSE) {
            y[2 * m] = wrap_scale(zr, 2);
            y[N2 - 1 - 2 * m] = wrap_scale(zi, -2);
        } else {
            // the turned sequence r of k_mdctq_unfold: r[2m] = re8, r[N/2 + 2m] = im8, odd entries mirror the even ones with
            // the sign flipped; x[i] = q15(r[N/4 + i], cof) for i < 3N/4, q15(-r[i - 3N/4], cof) behind
            const int re8 = wrap_scale(q15(zr, cof), 8), im8 = wrap_scale(q15(zi, cof), 8);
            auto put = [&](int ri, int val) {
                if (ri >= N4) y[ri - N4] = q15(val, cof);
                else y[ri + 3 * N4] = q15(wrap_neg(val), cof);
            };
            put(2 * m, re8);
            put(N - 1 - 2 * m, wrap_neg(re8));
            put(N2 + 2 * m, im8);
            put(N2 - 1 - 2 * m, wrap_neg(im8));
        }
    }
    __syncthreads();
    {
        int *dst = out + (size_t)tr0 * out_len;
        const int total = ntr * out_len;
        if (out_len == N) {
            for (int e = tid; e < total; e += FFT_THREADS) dst[e] = buf[e];
        } else {
            for (int e = tid; e < total; e += FFT_THREADS) dst[e] = buf[((e >> (log2n4 + 1)) << (log2n4 + 2)) + (e & (N2 - 1))];
        }
    }
}

// MDCT_FIXED_FFT whole (llz_mdct_fixed.c:155-196), the same way: modulate the frame into the N-point transform's LDS image
// (k_mdctq_modulate's expressions; the inverse continues the N/2 coefficients with odd symmetry), the radix-2 passes of the
// bit-exact Q15 transform -- forward for the MDCT, inverse (decimation in time from the bit-reversed image, then >> log2 N:
// llz_fft_fixed.c:187-215) for the IMDCT -- and k_mdctq_demodulate's rotation of the bins the direction keeps, row by row.
//   forward: in = x [count][N], out = X [count][N/2];   inverse: in = X [count][N/2], out = x [count][N]
template <bool INVERSE>
__global__ void __launch_bounds__(FFT_THREADS)
k_mdct1_q15(const int *__restrict__ in, int *__restrict__ out, int count, int N, int log2n, const short2 *__restrict__ pre,
            const short2 *__restrict__ post, const short *__restrict__ cs, int tpw, unsigned groups)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int N2 = N >> 1;
    const int tstride = fft_phys(N) + 1;
    cpx<int> *s = reinterpret_cast<cpx<int> *>(smem_raw);
    cpx<short> *tw = reinterpret_cast<cpx<short> *>(s + (size_t)tpw * tstride);
    const int tid = threadIdx.x;
    const int tr0 = blockIdx.x * tpw;
    const int ntr = min(tpw, count - tr0);
    const int in_len = INVERSE ? N2 : N, out_len = INVERSE ? N : N2;
    fft_load_twiddles(tw, cs, N, tid);
    for (int e = tid; e < ntr * N; e += FFT_THREADS) {
        const int tr = e >> log2n, k = e & (N - 1);
        const int *src = in + (size_t)(tr0 + tr) * in_len;
        int v;
        if (!INVERSE) v = src[k];
        else v = k < N2 ? src[k] : wrap_neg(src[N - 1 - k]);
        const short2 w = pre[k];
        cpx<int> z;
        z.re = q15(v, w.x);
        z.im = q15(v, w.y);
        // the inverse passes start from the bit-reversed image (llz_fft_fixed.c:187-195)
        const int at = INVERSE ? (int)(__brev((unsigned)k) >> (32 - log2n)) : k;
        s[tr * tstride + fft_phys(at)] = z;
    }
    __syncthreads();
    {
        int done = 0;
#pragma unroll 1
        for (int p = 0; p < 4; p++) {
            const int G = (groups >> (4 * p)) & 15;
            if (G == 0) break;
            const int log2step = INVERSE ? done : (log2n - done - G);
            switch (G) {
            case 1: fft_pass<arith_q15, 1, INVERSE>(s, ntr, N, log2n, log2step, tstride, tw, tid); break;
            case 2: fft_pass<arith_q15, 2, INVERSE>(s, ntr, N, log2n, log2step, tstride, tw, tid); break;
            case 3: fft_pass<arith_q15, 3, INVERSE>(s, ntr, N, log2n, log2step, tstride, tw, tid); break;
            default: fft_pass<arith_q15, 4, INVERSE>(s, ntr, N, log2n, log2step, tstride, tw, tid); break;
            }
            done += G;
        }
    }
    for (int e = tid; e < ntr * out_len; e += FFT_THREADS) {
        const int tr = e / out_len, k = e - tr * out_len;
        // forward: bin k sits at the bit-reversed position; inverse: sample k in place, scaled by 2^-log2n
        cpx<int> p = s[tr * tstride + fft_phys(INVERSE ? k : (int)(__brev((unsigned)k) >> (32 - log2n)))];
        if (INVERSE) {
            p.re = arith_q15::scale_out(p.re, log2n);
            p.im = arith_q15::scale_out(p.im, log2n);
        }
        const short2 w = post[k];
        const int d = wrap_sub(q15(p.re, w.x), q15(p.im, w.y));
        out[(size_t)(tr0 + tr) * out_len + k] = INVERSE ? (int)((unsigned)d << 1) : d;
    }
}

} // namespace

// MDCT_FIXED_FFT in one launch; N a power of two in 4..4096; pre: N (cos, sin) Q15 pairs; post: N/2 (forward) or N (inverse)
// pairs; cs: the N-point transform's table
extern "C" int llzs_mdct1_q15(const int *in, int *out, int count, int N, const short *pre, const short *post, const short *cs,
                              int inverse, void *stream)
{
    int log2n = 0;
    while ((1 << log2n) < N) log2n++;
    if (!in || !out || !pre || !post || !cs || count < 1 || N < 4 || N > 4096 || (1 << log2n) != N) {
        llzs_set_error("mdct1_q15: bad arguments (N=%d must be a power of two in 4..4096, count=%d)", N, count);
        return LLZ_ERR_ARG;
    }
    int tpw = 2048 / N;
    if (tpw < 1) tpw = 1;
    if (tpw > count) tpw = count;
    const int tstride = N + (N >> 5) + 1;
    const size_t lds = (size_t)tpw * tstride * 2 * sizeof(int) + (size_t)tw_entries(N) * 2 * sizeof(short);
    const unsigned blocks = (unsigned)((count + tpw - 1) / tpw);
    const short2 *p2 = reinterpret_cast<const short2 *>(pre), *q2 = reinterpret_cast<const short2 *>(post);
    if (inverse) {
        if (lds >= 64 * 1024)
            LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mdct1_q15<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mdct1_q15<true>, dim3(blocks), dim3(FFT_THREADS), lds, as_stream(stream), in, out, count, N, log2n,
                           p2, q2, cs, tpw, fft_groups(log2n));
    } else {
        if (lds >= 64 * 1024)
            LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mdct1_q15<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mdct1_q15<false>, dim3(blocks), dim3(FFT_THREADS), lds, as_stream(stream), in, out, count, N, log2n,
                           p2, q2, cs, tpw, fft_groups(log2n));
    }
    LLZ_LAUNCH_CHECK("k_mdct1_q15");
    return LLZ_OK;
}

// MDCT_FIXED_FFT4 in one launch; N a power of two in 8..16384; pre / post: N/4 (cos, sin) Q15 pairs; cs: the N/4-point
// transform's table
extern "C" int llzs_mdct4_q15(const int *in, int *out, int count, int N, const short *pre, const short *post,
                              const short *cs, int inverse, int cof, void *stream)
{
    int log2n = 0;
    while ((1 << log2n) < N) log2n++;
    if (!in || !out || !pre || !post || !cs || count < 1 || N < 8 || N > 16384 || (1 << log2n) != N) {
        llzs_set_error("mdct4_q15: bad arguments (N=%d must be a power of two in 8..16384, count=%d)", N, count);
        return LLZ_ERR_ARG;
    }
    const int N4 = N >> 2, log2n4 = log2n - 2;
    int tpw = 2048 / N4;
    if (tpw < 1) tpw = 1;
    if (tpw > count) tpw = count;
    const int tstride = N4 + (N4 >> 5) + 1;
    const int twe = tw_entries(N4) + (tw_entries(N4) & 1);
    const size_t lds = (size_t)tpw * tstride * 2 * sizeof(int) + (size_t)twe * 2 * sizeof(short) + (size_t)tpw * N * sizeof(int);
    const unsigned blocks = (unsigned)((count + tpw - 1) / tpw);
    const short2 *p2 = reinterpret_cast<const short2 *>(pre), *q2 = reinterpret_cast<const short2 *>(post);
    if (inverse) {
        if (lds >= 64 * 1024)
            LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mdct4_q15<true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mdct4_q15<true>, dim3(blocks), dim3(FFT_THREADS), lds, as_stream(stream), in, out, count, N,
                           log2n4, p2, q2, cs, tpw, fft_groups(log2n4), cof);
    } else {
        if (lds >= 64 * 1024)
            LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mdct4_q15<false>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mdct4_q15<false>, dim3(blocks), dim3(FFT_THREADS), lds, as_stream(stream), in, out, count, N,
                           log2n4, p2, q2, cs, tpw, fft_groups(log2n4), cof);
    }
    LLZ_LAUNCH_CHECK("k_mdct4_q15");
    return LLZ_OK;
}

extern "C" int llzs_mdctq_sums(const short *A, const int *x, int *y, int count, int rows, int cols, int quarter_over_n,
                               void *stream)
{
    if (!A || !x || !y || count < 1 || count > 65535 || rows < 1 || cols < 1) {
        llzs_set_error("mdctq_sums: bad arguments");
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_mdctq_sums, dim3((unsigned)((rows + 255) / 256), (unsigned)count), dim3(256), 0,
                       as_stream(stream), A, x, y, rows, cols, quarter_over_n);
    LLZ_LAUNCH_CHECK("k_mdctq_sums");
    return LLZ_OK;
}

extern "C" int llzs_mdctq_step(int quarter, int post, const int *in, int *out, const short *tw, int count, int N,
                               int inverse, int cof, void *stream)
{
    if (!in || !out || !tw || count < 1 || count > 65535 || N < 4 || (N & (N - 1))) {
        llzs_set_error("mdctq_step: bad arguments (N=%d count=%d)", N, count);
        return LLZ_ERR_ARG;
    }
    const short2 *t2 = reinterpret_cast<const short2 *>(tw);
    const dim3 grid((unsigned)((N + 255) / 256), (unsigned)count), block(256);
    if (!quarter && !post)
        hipLaunchKernelGGL(k_mdctq_modulate, grid, block, 0, as_stream(stream), in, out, t2, N, inverse);
    else if (!quarter)
        hipLaunchKernelGGL(k_mdctq_demodulate, grid, block, 0, as_stream(stream), in, out, t2, N, inverse);
    else if (!post)
        hipLaunchKernelGGL(k_mdctq_fold, grid, block, 0, as_stream(stream), in, out, t2, N, inverse);
    else
        hipLaunchKernelGGL(k_mdctq_unfold, grid, block, 0, as_stream(stream), in, out, t2, N, inverse, cof);
    LLZ_LAUNCH_CHECK("k_mdctq_step");
    return LLZ_OK;
}
