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
// A workgroup holds 2048 points in LDS (one 2048/4096-point transform or several smaller ones); the log2 N radix-2
// stages run as 1-3 passes of up to four stages fused in registers (16 elements per lane), a barrier between passes. Twiddle tables come from the host
// (never recomputed on the device: SURVEY.md H4/H5).
#include <stdlib.h>
#include <array>
#include <map>
#include <mutex>
#include "fft_core.hpp"
#include "fft32.hpp"

namespace {

// groups: up to four passes, G of pass p in bits [4p, 4p+4) (0 = no pass). tpw transforms per workgroup.
template <typename A, bool INVERSE>
__global__ void __launch_bounds__(FFT_THREADS)
k_fft_radix2(typename A::data_t *__restrict__ data, int count, int size, int log2n,
             const typename A::tw_t *__restrict__ cs /* size cos, then size sin */, int tpw, unsigned groups)
{
    typedef typename A::data_t T;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cpx<T> *s = reinterpret_cast<cpx<T> *>(smem_raw);
    const int tid = threadIdx.x;
    const int tr0 = blockIdx.x * tpw;
    const int ntr = min(tpw, count - tr0);                 // transforms this workgroup really has
    cpx<T> *g = reinterpret_cast<cpx<T> *>(data) + (size_t)tr0 * size;
    const int tstride = fft_phys(size) + 1;
    const int total = ntr << log2n;
    cpx<typename A::tw_t> *tw = reinterpret_cast<cpx<typename A::tw_t> *>(s + (size_t)tpw * tstride);
    fft_load_twiddles(tw, cs, size, tid);

    // load (inverse: through the bit-reversal, float flavour divides by N here: llz_fft.c:187-195)
    for (int e = tid; e < total; e += FFT_THREADS) {
        const int tr = e >> log2n, i = e & (size - 1);
        if (!INVERSE) {
            s[tr * tstride + fft_phys(i)] = g[e];
        } else {
            cpx<T> v = g[(tr << log2n) + (int)(__brev((unsigned)i) >> (32 - log2n))];
            v.re = A::scale_in(v.re, size, log2n);
            v.im = A::scale_in(v.im, size, log2n);
            s[tr * tstride + fft_phys(i)] = v;
        }
    }
    __syncthreads();

    int done = 0;                                          // stages finished so far
#pragma unroll 1
    for (int p = 0; p < 4; p++) {
        const int G = (groups >> (4 * p)) & 15;
        if (G == 0) break;
        // forward: first stage of the pass has half-span size >> (done+1), elements step = that >> (G-1)
        // inverse: first stage has half-span 1 << done = step
        const int log2step = INVERSE ? done : (log2n - done - G);
        switch (G) {
        case 1: fft_pass_any<A, 1, INVERSE>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        case 2: fft_pass_any<A, 2, INVERSE>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        case 3: fft_pass_any<A, 3, INVERSE>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        default: fft_pass_any<A, 4, INVERSE>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        }
        done += G;
    }

    // store (forward: through the bit-reversal gather, llz_fft.c:155-163; fixed inverse: >> log2 N, :212-215)
    for (int e = tid; e < total; e += FFT_THREADS) {
        const int tr = e >> log2n, i = e & (size - 1);
        if (!INVERSE) {
            g[e] = s[tr * tstride + fft_phys((int)(__brev((unsigned)i) >> (32 - log2n)))];
        } else {
            cpx<T> v = s[tr * tstride + fft_phys(i)];
            v.re = A::scale_out(v.re, log2n);
            v.im = A::scale_out(v.im, log2n);
            g[e] = v;
        }
    }
}


// FFT autocorrelation (reference libllzfilter/llz_corr.c:155-177) fused in LDS: real frame -> zero-padded complex ->
// forward passes (bins end up bit-reversed, which is exactly the order the inverse DIT passes consume) -> power
// spectrum of the first n bins, everything else zero, 1/F folded in -> inverse passes -> r[k] = 2 Re.  One read of the
// frame and p+1 floats written per frame instead of five launches over a 2F-float buffer.
__global__ void __launch_bounds__(FFT_THREADS)
k_acf_fused_f32(const float *__restrict__ x, float *__restrict__ r, int frames, int n, int p, int size, int log2n,
                const float *__restrict__ cs, int tpw, unsigned groups)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cpx<float> *s = reinterpret_cast<cpx<float> *>(smem_raw);
    const int tid = threadIdx.x;
    const int tr0 = blockIdx.x * tpw;
    const int ntr = min(tpw, frames - tr0);
    const int tstride = fft_phys(size) + 1;
    const int total = ntr << log2n;
    cpx<float> *tw = s + (size_t)tpw * tstride;
    fft_load_twiddles(tw, cs, size, tid);
    for (int e = tid; e < total; e += FFT_THREADS) {
        const int tr = e >> log2n, i = e & (size - 1);
        cpx<float> v;
        v.re = i < n ? x[(size_t)(tr0 + tr) * n + i] : 0.f;
        v.im = 0.f;
        s[tr * tstride + fft_phys(i)] = v;
    }
    __syncthreads();
    int done = 0;
#pragma unroll 1
    for (int pss = 0; pss < 4; pss++) {
        const int G = (groups >> (4 * pss)) & 15;
        if (G == 0) break;
        const int log2step = log2n - done - G;
        switch (G) {
        case 1: fft_pass_f32<1, false>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        case 2: fft_pass_f32<2, false>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        case 3: fft_pass_f32<3, false>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        default: fft_pass_f32<4, false>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        }
        done += G;
    }
    // position j holds bin brev(j): keep |X|^2 / F for bins < n (llz_corr.c:165-170; the 1/F of llz_ifft folded in)
    const float inv = 1.0f / (float)size;
    for (int e = tid; e < total; e += FFT_THREADS) {
        const int tr = e >> log2n, j = e & (size - 1);
        const int bin = (int)(__brev((unsigned)j) >> (32 - log2n));
        cpx<float> &v = s[tr * tstride + fft_phys(j)];
        const float pw = bin < n ? __builtin_fmaf(v.re, v.re, v.im * v.im) * inv : 0.f;
        v.re = pw;
        v.im = 0.f;
    }
    __syncthreads();
    done = 0;
#pragma unroll 1
    for (int pss = 0; pss < 4; pss++) {
        const int G = (groups >> (4 * pss)) & 15;
        if (G == 0) break;
        switch (G) {
        case 1: fft_pass_f32<1, true>(s, ntr, size, log2n, done, tstride, tw, tid); break;
        case 2: fft_pass_f32<2, true>(s, ntr, size, log2n, done, tstride, tw, tid); break;
        case 3: fft_pass_f32<3, true>(s, ntr, size, log2n, done, tstride, tw, tid); break;
        default: fft_pass_f32<4, true>(s, ntr, size, log2n, done, tstride, tw, tid); break;
        }
        done += G;
    }
    for (int e = tid; e < ntr * (p + 1); e += FFT_THREADS) {
        const int tr = e / (p + 1), k = e - tr * (p + 1);
        r[(size_t)(tr0 + tr) * (p + 1) + k] = s[tr * tstride + fft_phys(k)].re * 2.f;      // llz_corr.c:173
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Windowed-FFT analysis / synthesis frames (reference libllzfilter/llz_asmodel.c:180-310), many
// channels and frames per launch.  size = R * frame_len with R = 4 (3/4 overlap) or 2 (1/2 overlap).

// analysis: frame f of channel c is samples [(f+1)F - size, (f+1)F) of concat(hist, x) times the window; bins 0..size/2
// of its transform go to re/im[(c*frames + f)*bins + b] (llz_asmodel.c:188-204).  tpw frames share a workgroup.
__global__ void __launch_bounds__(FFT_THREADS)
k_stft_analysis_f32(const float *__restrict__ x, const float *__restrict__ hist, float *__restrict__ re,
                    float *__restrict__ im, const float *__restrict__ w, int frames, int F, int size, int log2n,
                    const float *__restrict__ cs, int tpw, unsigned groups, long x_pitch, long total_tr)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cpx<float> *s = reinterpret_cast<cpx<float> *>(smem_raw);
    const int tid = threadIdx.x;
    const long tr0 = (long)blockIdx.x * tpw;
    const int ntr = (int)min((long)tpw, total_tr - tr0);
    const int tstride = fft_phys(size) + 1;
    const int total = ntr << log2n;
    const int keep = size - F;                                         // history samples in front of a call
    cpx<float> *tw = s + (size_t)tpw * tstride;
    fft_load_twiddles(tw, cs, size, tid);
    for (int e = tid; e < total; e += FFT_THREADS) {
        const int tr = e >> log2n, i = e & (size - 1);
        const long g = tr0 + tr;
        const int c = (int)(g / frames), f = (int)(g - (long)c * frames);
        const long t = (long)(f + 1) * F - size + i;                   // sample index inside this call
        const float v = t >= 0 ? x[(size_t)c * x_pitch + t] : hist[(size_t)c * keep + (keep + t)];
        cpx<float> z;
        z.re = v * w[i];
        z.im = 0.f;
        s[tr * tstride + fft_phys(i)] = z;
    }
    __syncthreads();
    int done = 0;
#pragma unroll 1
    for (int pss = 0; pss < 4; pss++) {
        const int G = (groups >> (4 * pss)) & 15;
        if (G == 0) break;
        const int log2step = log2n - done - G;
        switch (G) {
        case 1: fft_pass_f32<1, false>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        case 2: fft_pass_f32<2, false>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        case 3: fft_pass_f32<3, false>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        default: fft_pass_f32<4, false>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        }
        done += G;
    }
    const int bins = (size >> 1) + 1;                                  // position j holds bin brev(j)
    for (int tr = 0; tr < ntr; tr++) {
        const size_t o = (size_t)(tr0 + tr) * bins;
        for (int b = tid; b < bins; b += FFT_THREADS) {
            const cpx<float> v = s[tr * tstride + fft_phys((int)(__brev((unsigned)b) >> (32 - log2n)))];
            re[o + b] = v.re;
            im[o + b] = v.im;
        }
    }
}

// fft_len = 1024 analysis frames on the half-wave machinery of k_fft1024_f32: the windowed samples go from HBM straight
// into registers (imaginary parts zero), bins 0..512 straight back; 8 frames per workgroup.
__global__ void __launch_bounds__(256)
k_stft_analysis1024_f32(const float *__restrict__ x, const float *__restrict__ hist, float *__restrict__ re,
                        float *__restrict__ im, const float *__restrict__ w, int frames, int F,
                        const float *__restrict__ cs, long x_pitch, long total_tr)
{
    __shared__ float2 s_tw[1024];
    __shared__ float bufs[8][OLS_XBUF];
    const int tid = threadIdx.x, hw = tid >> 5, l5 = tid & 31;
    for (int i = tid; i < 1024; i += 256) {
        const int m = ((i >> 5) * (i & 31)) & 1023;
        s_tw[i] = make_float2(cs[m], -cs[1024 + m]);
    }
    __syncthreads();
    const long g = (long)blockIdx.x * 8 + hw;
    if (g >= total_tr) return;
    const int c = (int)(g / frames), f = (int)(g - (long)c * frames);
    const int keep = 1024 - F;
    const long t0 = (long)(f + 1) * F - 1024;
    const float *row = x + (size_t)c * x_pitch;
    const float *hrow = hist + (size_t)c * keep;
    cf v[32];
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const int i = l5 + 32 * j;
        const long t = t0 + i;
        const float smp = t >= 0 ? row[t] : hrow[keep + t];
        v[j] = cf{smp * w[i], 0.f};
    }
    fft32<false>(v);
    transpose_twiddle<false>(v, bufs[hw], s_tw, l5);
    fft32<false>(v);
    const size_t o = (size_t)g * 513;
#pragma unroll
    for (int q = 0; q < 32; q++) {
        const int bin = l5 + 32 * brev5(q);                         // v[q] = X[l5 + 32 brev5(q)]
        if (bin <= 512) {
            re[o + bin] = v[q].x;
            im[o + bin] = v[q].y;
        }
    }
}

// synthesis: a workgroup owns output blocks [b0, b1) of one channel.  Block t (frame_len samples) is the sum of the
// windowed inverse transforms of frames t-R+1 .. t (llz_asmodel.c:279-304), so the workgroup walks frames
// max(0, b0-R+1) .. b1-1 in groups of tpw, keeps the running overlap-add tail (size - F samples) in LDS and drops the
// blocks in front of b0 (their sums are incomplete; the first run of a channel starts from the handle's tail instead).
// Accumulation order per sample is the reference's: oldest frame first.
__global__ void __launch_bounds__(FFT_THREADS)
k_stft_synthesis_f32(const float *__restrict__ re, const float *__restrict__ im, float *__restrict__ x,
                     const float *__restrict__ ola_old, float *__restrict__ ola_new, const float *__restrict__ w,
                     int frames, int F, int size, int log2n, const float *__restrict__ cs, int tpw, unsigned groups,
                     long x_pitch, int run_len, int runs, float magic)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tstride = fft_phys(size) + 1;
    cpx<float> *s = reinterpret_cast<cpx<float> *>(smem_raw);
    cpx<float> *tw = s + (size_t)tpw * tstride;
    float *carry = reinterpret_cast<float *>(tw + tw_entries(size));      // size - F floats
    const int tid = threadIdx.x;
    const int c = blockIdx.x / runs, run = blockIdx.x - c * runs;
    const int b0 = run * run_len, b1 = min(frames, b0 + run_len);
    const int R = size / F, keep = size - F, bins = (size >> 1) + 1;
    const int fs = max(0, b0 - (R - 1));
    fft_load_twiddles(tw, cs, size, tid);
    for (int q = tid; q < keep; q += FFT_THREADS) carry[q] = fs == 0 ? ola_old[(size_t)c * keep + q] : 0.f;
    const float inv = 1.0f / (float)size;                                  // llz_ifft divides by N (llz_fft.c:187-195)
    for (int g0 = fs; g0 < b1; g0 += tpw) {
        const int ng = min(tpw, b1 - g0);
        __syncthreads();                                                   // carry written, s free
        // spectra into bit-reversed positions: bins 0..size/2 as given, the upper half by Hermitian symmetry
        for (int tr = 0; tr < ng; tr++) {
            const size_t o = ((size_t)c * frames + g0 + tr) * bins;
            for (int b = tid; b < bins; b += FFT_THREADS) {
                cpx<float> v;
                v.re = re[o + b] * inv;
                v.im = im[o + b] * inv;
                s[tr * tstride + fft_phys((int)(__brev((unsigned)b) >> (32 - log2n)))] = v;
                if (b > 0 && b < (size >> 1)) {
                    v.im = -v.im;
                    s[tr * tstride + fft_phys((int)(__brev((unsigned)(size - b)) >> (32 - log2n)))] = v;
                }
            }
        }
        __syncthreads();
        int done = 0;
#pragma unroll 1
        for (int pss = 0; pss < 4; pss++) {
            const int G = (groups >> (4 * pss)) & 15;
            if (G == 0) break;
            switch (G) {
            case 1: fft_pass_f32<1, true>(s, ng, size, log2n, done, tstride, tw, tid); break;
            case 2: fft_pass_f32<2, true>(s, ng, size, log2n, done, tstride, tw, tid); break;
            case 3: fft_pass_f32<3, true>(s, ng, size, log2n, done, tstride, tw, tid); break;
            default: fft_pass_f32<4, true>(s, ng, size, log2n, done, tstride, tw, tid); break;
            }
            done += G;
        }
        // overlap-add over the group's span: position p counts from the group's first block
        const int span = (ng - 1) * F + size;                              // <= 2048
        float acc[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int p = tid + m * FFT_THREADS;
            float a = 0.f;
            if (p < span) {
                a = p < keep ? carry[p] : 0.f;
                const int k_hi = min(ng - 1, p / F);                       // frames k with 0 <= p - kF < size
                const int k_lo = p < size ? 0 : (p - size) / F + 1;
                for (int k = k_lo; k <= k_hi; k++) {
                    const int i = p - k * F;
                    a += s[k * tstride + fft_phys(i)].re * w[i];
                }
            }
            acc[m] = a;
        }
        __syncthreads();                                                   // every read of carry and s is done
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int p = tid + m * FFT_THREADS;
            if (p < span) {
                if (p < ng * F) {
                    if (g0 + p / F >= b0) x[(size_t)c * x_pitch + (size_t)g0 * F + p] = magic * acc[m];
                } else {
                    carry[p - ng * F] = acc[m];
                }
            }
        }
    }
    __syncthreads();
    if (b1 == frames)
        for (int q = tid; q < keep; q += FFT_THREADS) ola_new[(size_t)c * keep + q] = carry[q];
}

// 1024-point float32 transforms, the overlap-save size: one HALF-WAVE per transform, two in-register 32-point passes and
// one LDS transpose between them (fft32.hpp, the machinery of K4).  Input goes from HBM straight into registers (lane l
// takes x[l + 32 j]: 256-byte runs) and the result straight back (X[l + 32 k2]), so a transform costs ~600 vector
// instructions per lane pair instead of the ~1550 of the staged radix-2 passes, which are bound by instruction issue.
template <bool INV>
__global__ void __launch_bounds__(256)
k_fft1024_f32(float *__restrict__ data, int count, const float *__restrict__ cs /* 1024 cos, then 1024 sin */)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ float2 s_tw[1024];                                  // W_1024^(a*b) = exp(-2 pi j a b / 1024), [a][b]
    __shared__ float bufs[8][OLS_XBUF];
    const int tid = threadIdx.x, hw = tid >> 5, l5 = tid & 31;
    for (int i = tid; i < 1024; i += 256) {
        const int m = ((i >> 5) * (i & 31)) & 1023;
        s_tw[i] = make_float2(cs[m], -cs[1024 + m]);
    }
    __syncthreads();
    const long t = (long)blockIdx.x * 8 + hw;
    if (t >= count) return;
    f32x2 *g = reinterpret_cast<f32x2 *>(data) + t * 1024;
    cf v[32];
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const f32x2 x = __builtin_nontemporal_load(&g[l5 + 32 * j]);
        // the inverse divides by N on the way in, as llz_ifft does (llz_fft.c:187-195); 1/1024 is exact
        v[j] = INV ? cf{x.x * (1.0f / 1024.0f), x.y * (1.0f / 1024.0f)} : cf{x.x, x.y};
    }
    fft32<INV>(v);                                                 // over j: v[q] = Y[k1 = brev5(q)] of column l5
    transpose_twiddle<INV>(v, bufs[hw], s_tw, l5);                 // * W^(k1 * l5), then lane k1 holds row k1
    fft32<INV>(v);                                                 // over the column index: v[q] = X[l5 + 32 brev5(q)]
#pragma unroll
    for (int q = 0; q < 32; q++)
        __builtin_nontemporal_store((f32x2){v[q].x, v[q].y}, &g[l5 + 32 * brev5(q)]);
}

// FFT autocorrelation for fft_len = 2048 (frames of 513..1024 samples) on the half-wave machinery.  Both 2048-point
// transforms of llz_corr.c:155-177 act on real data, so each is ONE 1024-point complex transform:
//   forward: z[m] = x[2m] + j x[2m+1]; Z = FFT_1024(z); with Zm = Z[1024-k]: Xe = (Z[k] + conj(Zm))/2,
//            Xo = (Z[k] - conj(Zm))/(2j), T = W_2048^k Xo:  X[k] = Xe + T,  X[1024-k] = conj(Xe - T);
//   power:   P[b] = |X[b]|^2 / 2048 for b < n (the reference squares only the first n bins), else 0;
//   inverse: r[k] = 2 Re sum_{b<n} P[b] W^-bk is the real inverse transform of the symmetric spectrum S[b] = S[2048-b] =
//            P[b] (b >= 1), S[0] = 2 P[0], S[1024] = 0:  G[b] = (S[b] + S[1024-b]) + j conj(W^b) (S[b] - S[1024-b]),
//            g = IFFT_1024(G) unnormalised, r[2m] = Re g[m], r[2m+1] = Im g[m].
// The mirrored bin lives in lane (32 - l) of the same half-wave: one more LDS round trip per plane.  For p < 64 only
// g[0..31] is needed, i.e. bin 0 of the second register pass: 31 complex adds instead of a 32-point transform.
// (184 VGPRs as written: three waves per SIMD are asked for, 168 registers and a few spilled -- 0.73 -> 0.60 ms)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
k_acf2048_f32(const float *__restrict__ x, float *__restrict__ r, int frames, int n, int p,
              const float *__restrict__ cs /* 2048 cos, then 2048 sin of 2 pi i / 2048 */)
{
    __shared__ float2 s_tw[1024];                                  // W_1024^(a*b), [a][b]
    __shared__ float2 s_w2[1024];                                  // W_2048^k, k < 1024
    __shared__ float bufs[8][OLS_XBUF];
    const int tid = threadIdx.x, hw = tid >> 5, l5 = tid & 31;
    for (int i = tid; i < 1024; i += 256) {
        const int m = (2 * (i >> 5) * (i & 31)) & 2047;
        s_tw[i] = make_float2(cs[m], -cs[2048 + m]);
        s_w2[i] = make_float2(cs[i], -cs[2048 + i]);
    }
    __syncthreads();
    const long t = (long)blockIdx.x * 8 + hw;
    if (t >= frames) return;
    const float *g = x + t * n;
    float *buf = bufs[hw];
    cf v[32];
#pragma unroll
    for (int j = 0; j < 16; j++) {                                 // 2 m < 1024: the upper half of z is padding
        const int i0 = 2 * (l5 + 32 * j);
        v[j].x = i0 < n ? g[i0] : 0.f;
        v[j].y = i0 + 1 < n ? g[i0 + 1] : 0.f;
    }
#pragma unroll
    for (int j = 16; j < 32; j++) v[j] = cf{0.f, 0.f};
    fft32<false>(v);
    transpose_twiddle<false>(v, buf, s_tw, l5);
    fft32<false>(v);                                               // v[q] = Z[l5 + 32 brev5(q)]
    // mirrored bins: Z[1024 - k] sits in lane (32 - l5) & 31 at column index 31 - j (lane 0: (32 - j) & 31)
    const int lm = (32 - l5) & 31;
    float mx[32];
#pragma unroll
    for (int q = 0; q < 32; q++) buf[xaddr(brev5(q), l5)] = v[q].x;
    OLS_WAVE_SYNC();
#pragma unroll
    for (int q = 0; q < 32; q++) {
        const int j = brev5(q);
        mx[q] = buf[l5 ? xaddr(31 - j, lm) : xaddr((32 - j) & 31, 0)];
    }
    OLS_WAVE_SYNC();
#pragma unroll
    for (int q = 0; q < 32; q++) buf[xaddr(brev5(q), l5)] = v[q].y;
    OLS_WAVE_SYNC();
    const float sc = 1.0f / (4.0f * 2048.0f);                      // the two halvings of (Xe, Xo) and llz_ifft's 1/N
#pragma unroll
    for (int q = 0; q < 32; q++) {
        const int j = brev5(q);
        const int k = l5 + 32 * j;
        const float my = buf[l5 ? xaddr(31 - j, lm) : xaddr((32 - j) & 31, 0)];
        const float2 w = s_w2[k];                                  // (cos, -sin) of pi k / 1024
        const cf xe = {v[q].x + mx[q], v[q].y - my};               // 2 Xe
        const cf xo = {v[q].y + my, mx[q] - v[q].x};               // 2 Xo
        const cf T = cmul<false>(xo, cf{w.x, w.y});
        const cf a = cadd(xe, T), b = csub(xe, T);
        float sk = __builtin_fmaf(a.x, a.x, a.y * a.y) * sc, sm = __builtin_fmaf(b.x, b.x, b.y * b.y) * sc;
        if (k >= n) sk = 0.f;
        if (1024 - k >= n) sm = 0.f;
        if (k == 0) { sk *= 2.f; sm = 0.f; }                       // S[0] = 2 P[0]; the mirror of bin 0 is bin 1024: unused
        const float dk = sk - sm;
        v[q] = cf{__builtin_fmaf(w.y, dk, sk + sm), w.x * dk};     // (S + Sm) + j (c + j s) dk,  w.y = -s
    }
    OLS_WAVE_SYNC();
    cf u[32];
#pragma unroll
    for (int j = 0; j < 32; j++) u[j] = v[brev5(j)];               // bin order -> natural order: register renaming
    fft32<true>(u);
    transpose_twiddle<true>(u, buf, s_tw, l5);
    float *rr = r + t * (p + 1);
    if (p < 64) {                                                  // only g[l5] = the sum over the column index
        cf acc = u[0];
#pragma unroll
        for (int q = 1; q < 32; q++) acc = cadd(acc, u[q]);
        if (2 * l5 <= p) rr[2 * l5] = acc.x;
        if (2 * l5 + 1 <= p) rr[2 * l5 + 1] = acc.y;
    } else {
        fft32<true>(u);                                            // u[q] = g[l5 + 32 brev5(q)]
#pragma unroll
        for (int q = 0; q < 32; q++) {
            const int m = l5 + 32 * brev5(q);
            if (2 * m <= p) rr[2 * m] = u[q].x;
            if (2 * m + 1 <= p) rr[2 * m + 1] = u[q].y;
        }
    }
}

// FFT autocorrelation for fft_len = 4096 (frames of 1025..2048 samples): the real-input scheme of k_acf2048_f32 one size up --
// both 4096-point transforms are ONE 2048-point complex transform each, and that transform runs on a WHOLE WAVE as in
// fir_ols.hip (k_fir_ols2k_chain_f32): one radix-2 step splits it over the two half-waves, each of which runs the 1024-point
// machinery.  z[m] = x[2m] + j x[2m+1], m < 2048, and the frame is at most 2048 samples, so z[m] = 0 for m >= 1024:
//   forward (decimation in frequency):  Z[2k']   = FFT_1024( z[m] )            -> lower half-wave
//                                       Z[2k'+1] = FFT_1024( z[m] W_2048^m )   -> upper half-wave          (m < 1024)
//   a bin's mirror Z[2048 - k] has the parity of k, so it sits in the SAME half-wave: index (1024 - k') mod 1024 among the
//   even bins, 1023 - k' among the odd ones -- one LDS round trip per plane, as in k_acf2048_f32;
//   X[k] = Xe + W_4096^k Xo, power, symmetric spectrum and G[k] as there (4096 for 2048, 2048 for 1024);
//   inverse (decimation in time):  g[m], g[m + 1024] = S'[m] +- W_2048^-m D'[m],  S' / D' = IFFT_1024 of G's even / odd bins;
//   r[2m] = Re g[m], r[2m + 1] = Im g[m].
// Lane (half h, l5) owns the rows of parity h of the 64 x 32 sample block (row 2p + h, p < 32) at column l5: rows p and p + 16
// are 1024 samples apart, so the butterflies of both radix-2 steps are in-lane and one v_permlane32_swap per register pair
// sorts sums / differences to the lower / upper half-wave.
__global__ void __launch_bounds__(256, 2)
k_acf4096_f32(const float *__restrict__ x, float *__restrict__ r, int frames, int n, int p,
              const float *__restrict__ cs /* 4096 cos, then 4096 sin of 2 pi i / 4096 */)
{
    __shared__ float2 s_tw[1024];                                  // W_1024^(a*b), [a][b]
    __shared__ float2 s_w2[1024];                                  // W_2048^m, m < 1024
    __shared__ float2 s_w4[2048];                                  // W_4096^k, k < 2048
    __shared__ float bufs[8][OLS_XBUF];
    const int tid = threadIdx.x, l5 = tid & 31, half = (tid >> 5) & 1;
    for (int i = tid; i < 1024; i += 256) {
        const int m = (4 * (i >> 5) * (i & 31)) & 4095;
        s_tw[i] = make_float2(cs[m], -cs[4096 + m]);
        s_w2[i] = make_float2(cs[2 * i], -cs[4096 + 2 * i]);
        s_w4[i] = make_float2(cs[i], -cs[4096 + i]);
        s_w4[1024 + i] = make_float2(cs[1024 + i], -cs[4096 + 1024 + i]);
    }
    __syncthreads();
    const long t = (long)blockIdx.x * 4 + (tid >> 6);              // a wave per frame
    if (t >= frames) return;
    const float *g = x + t * n;
    float *buf = bufs[tid >> 5];
    const int rowoff = 32 * half + l5;
    // ---- this lane's rows of z (m = 64 q + rowoff < 1024; the upper half of z is padding) and the radix-2 step down
    cf w[32];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int i0 = 2 * (64 * q + rowoff);
        cf sm = {i0 < n ? g[i0] : 0.f, i0 + 1 < n ? g[i0 + 1] : 0.f};
        const float2 tw = s_w2[64 * q + rowoff];
        cf df = cmul<false>(sm, cf{tw.x, tw.y});
        swap32(sm.x, df.x);
        swap32(sm.y, df.y);
        w[2 * q] = sm;                                             // lower: z rows 2q, 2q+1; upper: the twiddled copies
        w[2 * q + 1] = df;
    }
    fft32<false>(w);
    transpose_twiddle<false>(w, buf, s_tw, l5);
    fft32<false>(w);                                               // w[q] = Z[2 k' + half], k' = l5 + 32 brev5(q)
    // ---- mirrored bins: even bins k' -> (1024 - k') mod 1024: lane (32 - l5) & 31, column 31 - j (lane 0: (32 - j) & 31);
    //      odd bins k' -> 1023 - k': lane 31 - l5, column 31 - j
    const int lm = half ? 31 - l5 : (32 - l5) & 31;
    const bool wrap = !half && l5 == 0;
    float mx[32];
#pragma unroll
    for (int q = 0; q < 32; q++) buf[xaddr(brev5(q), l5)] = w[q].x;
    OLS_WAVE_SYNC();
#pragma unroll
    for (int q = 0; q < 32; q++) {
        const int j = brev5(q);
        mx[q] = buf[wrap ? xaddr((32 - j) & 31, 0) : xaddr(31 - j, lm)];
    }
    OLS_WAVE_SYNC();
#pragma unroll
    for (int q = 0; q < 32; q++) buf[xaddr(brev5(q), l5)] = w[q].y;
    OLS_WAVE_SYNC();
    const float sc = 1.0f / (4.0f * 4096.0f);                      // the two halvings of (Xe, Xo) and llz_ifft's 1/N
#pragma unroll
    for (int q = 0; q < 32; q++) {
        const int j = brev5(q);
        const int k = 2 * (l5 + 32 * j) + half;                    // this lane's bin, k < 2048
        const float my = buf[wrap ? xaddr((32 - j) & 31, 0) : xaddr(31 - j, lm)];
        const float2 tw = s_w4[k];                                 // (cos, -sin) of 2 pi k / 4096
        const cf xe = {w[q].x + mx[q], w[q].y - my};               // 2 Xe
        const cf xo = {w[q].y + my, mx[q] - w[q].x};               // 2 Xo
        const cf T = cmul<false>(xo, cf{tw.x, tw.y});
        const cf a = cadd(xe, T), b = csub(xe, T);
        float sk = __builtin_fmaf(a.x, a.x, a.y * a.y) * sc, sm = __builtin_fmaf(b.x, b.x, b.y * b.y) * sc;
        if (k >= n) sk = 0.f;
        if (2048 - k >= n) sm = 0.f;
        if (k == 0) { sk *= 2.f; sm = 0.f; }                       // S[0] = 2 P[0]; the mirror of bin 0 is bin 2048: unused
        const float dk = sk - sm;
        w[q] = cf{__builtin_fmaf(tw.y, dk, sk + sm), tw.x * dk};   // (S + Sm) + j conj(W^k) dk
    }
    OLS_WAVE_SYNC();
    // ---- inverse transforms of the even / odd bins, then the radix-2 step up
    cf u[32];
#pragma unroll
    for (int j = 0; j < 32; j++) u[j] = w[brev5(j)];               // bin order -> natural order: register renaming
    fft32<true>(u);
    transpose_twiddle<true>(u, buf, s_tw, l5);
    fft32<true>(u);                                                // u[q] = S' / D' [32 brev5(q) + l5]
    float *rr = r + t * (p + 1);
#pragma unroll
    for (int q = 0; q < 16; q++) {
        cf P = u[brev5(2 * q)], Q = u[brev5(2 * q + 1)];
        swap32(P.x, Q.x);
        swap32(P.y, Q.y);                                          // lane (half, l5): P = S', Q = D' at m = 64 q + rowoff
        const float2 tw = s_w2[64 * q + rowoff];
        Q = cmul<true>(Q, cf{tw.x, tw.y});
        const cf lo = cadd(P, Q), hi = csub(P, Q);
        const int m = 64 * q + rowoff;
        if (2 * m <= p) rr[2 * m] = lo.x;
        if (2 * m + 1 <= p) rr[2 * m + 1] = lo.y;
        if (2 * (m + 1024) <= p) rr[2 * (m + 1024)] = hi.x;
        if (2 * (m + 1024) + 1 <= p) rr[2 * (m + 1024) + 1] = hi.y;
    }
}

// FFT autocorrelation for fft_len = 1024 (frames of 257..512 samples): the two transforms of llz_corr.c:155-177 as they
// stand (complex, zero imaginary parts) on the half-wave machinery -- twice the arithmetic of the real-input form above,
// but a half size of 512 = 2 x 16^2 has no single-group register transform with the mirrored bins in reach, and even so
// this is several times the staged kernel.  For p < 32 the second inverse pass is its bin 0 only.
__global__ void __launch_bounds__(256)
k_acf1024_f32(const float *__restrict__ x, float *__restrict__ r, int frames, int n, int p,
              const float *__restrict__ cs /* 1024 cos, then 1024 sin of 2 pi i / 1024 */)
{
    __shared__ float2 s_tw[1024];                                  // W_1024^(a*b), [a][b]
    __shared__ float bufs[8][OLS_XBUF];
    const int tid = threadIdx.x, hw = tid >> 5, l5 = tid & 31;
    for (int i = tid; i < 1024; i += 256) {
        const int m = ((i >> 5) * (i & 31)) & 1023;
        s_tw[i] = make_float2(cs[m], -cs[1024 + m]);
    }
    __syncthreads();
    const long t = (long)blockIdx.x * 8 + hw;
    if (t >= frames) return;
    const float *g = x + t * n;
    float *buf = bufs[hw];
    cf v[32];
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const int i = l5 + 32 * j;
        v[j] = cf{(j < 16 && i < n) ? g[i] : 0.f, 0.f};            // n <= 512: the upper half is padding
    }
    fft32<false>(v);
    transpose_twiddle<false>(v, buf, s_tw, l5);
    fft32<false>(v);                                               // v[q] = X[l5 + 32 brev5(q)]
    cf u[32];
#pragma unroll
    for (int j = 0; j < 32; j++) {                                 // bin l5 + 32 j sits in register brev5(j)
        const cf z = v[brev5(j)];
        const int bin = l5 + 32 * j;
        // |X|^2 / F of the first n bins, everything else zero (llz_corr.c:165-170; the 1/F of llz_ifft folded in)
        u[j] = cf{bin < n ? __builtin_fmaf(z.x, z.x, z.y * z.y) * (1.0f / 1024.0f) : 0.f, 0.f};
    }
    fft32<true>(u);
    transpose_twiddle<true>(u, buf, s_tw, l5);
    float *rr = r + t * (p + 1);
    if (p < 32) {                                                  // only g[l5] = the sum over the column index
        float acc = u[0].x;
#pragma unroll
        for (int q = 1; q < 32; q++) acc += u[q].x;
        if (l5 <= p) rr[l5] = 2.f * acc;                           // llz_corr.c:173
    } else {
        fft32<true>(u);                                            // u[q] = g[l5 + 32 brev5(q)]
#pragma unroll
        for (int q = 0; q < 32; q++) {
            const int k = l5 + 32 * brev5(q);
            if (k <= p) rr[k] = 2.f * u[q].x;
        }
    }
}

// fft_len = 1024 synthesis: as k_stft_synthesis_f32, with the inverse transforms on the half-wave machinery (8 frames per
// group, one per half-wave): bins from HBM straight into registers with the Hermitian upper half taken from the mirrored
// bin, windowed real output written to an LDS segment image, then the same overlap-add walk.
__global__ void __launch_bounds__(256)
k_stft_synthesis1024_f32(const float *__restrict__ re, const float *__restrict__ im, float *__restrict__ x,
                         const float *__restrict__ ola_old, float *__restrict__ ola_new, const float *__restrict__ w,
                         int frames, int F, const float *__restrict__ cs, long x_pitch, int run_len, int runs, float magic)
{
    constexpr int N = 1024, TPW = 8;
    __shared__ float2 s_tw[1024];
    __shared__ float bufs[TPW][OLS_XBUF];
    __shared__ float seg[TPW][N];
    __shared__ float carry[N];                                         // N - F used
    const int tid = threadIdx.x, hw = tid >> 5, l5 = tid & 31;
    for (int i = tid; i < 1024; i += 256) {
        const int m = ((i >> 5) * (i & 31)) & 1023;
        s_tw[i] = make_float2(cs[m], -cs[1024 + m]);
    }
    const int c = blockIdx.x / runs, run = blockIdx.x - c * runs;
    const int b0 = run * run_len, b1 = min(frames, b0 + run_len);
    const int R = N / F, keep = N - F;
    const int fs = max(0, b0 - (R - 1));
    for (int q = tid; q < keep; q += 256) carry[q] = fs == 0 ? ola_old[(size_t)c * keep + q] : 0.f;
    for (int g0 = fs; g0 < b1; g0 += TPW) {
        const int ng = min(TPW, b1 - g0);
        __syncthreads();                                               // carry and seg of the previous group are consumed
        if (hw < ng) {
            const size_t o = ((size_t)c * frames + g0 + hw) * 513;
            cf v[32];
#pragma unroll
            for (int j = 0; j < 32; j++) {
                const int k = l5 + 32 * j;
                const int kk = k <= 512 ? k : N - k;                   // upper half: conjugate of the mirrored bin
                const float a = re[o + kk] * (1.0f / 1024.0f), b = im[o + kk] * (1.0f / 1024.0f);
                v[j] = cf{a, k <= 512 ? b : -b};
            }
            fft32<true>(v);
            transpose_twiddle<true>(v, bufs[hw], s_tw, l5);
            fft32<true>(v);
#pragma unroll
            for (int q = 0; q < 32; q++) {
                const int n = l5 + 32 * brev5(q);                      // v[q].x = real part of sample n
                seg[hw][n] = v[q].x * w[n];
            }
        }
        __syncthreads();
        const int span = (ng - 1) * F + N;                             // <= 7*512 + 1024
        float acc[18];
#pragma unroll
        for (int m = 0; m < 18; m++) {
            const int p = tid + m * 256;
            float a = 0.f;
            if (p < span) {
                a = p < keep ? carry[p] : 0.f;
                const int k_hi = min(ng - 1, p / F);
                const int k_lo = p < N ? 0 : (p - N) / F + 1;
                for (int k = k_lo; k <= k_hi; k++) a += seg[k][p - k * F];
            }
            acc[m] = a;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 18; m++) {
            const int p = tid + m * 256;
            if (p < span) {
                if (p < ng * F) {
                    if (g0 + p / F >= b0) x[(size_t)c * x_pitch + (size_t)g0 * F + p] = magic * acc[m];
                } else {
                    carry[p - ng * F] = acc[m];
                }
            }
        }
    }
    __syncthreads();
    if (b1 == frames)
        for (int q = tid; q < keep; q += 256) ola_new[(size_t)c * keep + q] = carry[q];
}

// ---------------------------------------------------------------------------------------------------------------------
// Square sizes N = E*E beside 1024 (E = 16: N = 256, E = 64: N = 4096), float32: the same two-register-pass scheme as
// k_fft1024_f32 with a group of E lanes per transform (a quarter wave / a whole wave), E elements per lane.
// The E-point transform is the decimation-in-time form of fft32.hpp written for any E <= 64 (constants in 64ths of a
// turn); the inter-pass twiddles W_N^(k1*l) come from a [E][E] table laid out so that a group reads a contiguous row.
__device__ constexpr float kCos64[17] = {1.00000000000000000000f, 0.99518472667219692873f, 0.98078528040323043058f, 0.95694033573220882438f, 0.92387953251128673848f, 0.88192126434835504956f, 0.83146961230254523567f, 0.77301045336273699338f, 0.70710678118654757274f, 0.63439328416364548779f, 0.55557023301960228867f, 0.47139673682599780857f, 0.38268343236508983729f, 0.29028467725446233105f, 0.19509032201612833135f, 0.09801714032956077016f, 0.00000000000000006123f};

template <int E>
__device__ constexpr int brevE(int r)
{
    int o = 0;
    for (int b = 1, t = E >> 1; b < E; b <<= 1, t >>= 1)
        if (r & b) o |= t;
    return o;
}

// (a, b) -> (a + w b, a - w b), w = W_E^q (conjugated for the inverse), 0 <= q < E/2, Linzer-Feig form as in fft32.hpp
template <int E, bool INV>
__device__ __forceinline__ void bfly_ditE(cf &a, cf &b, int q)
{
    const int q64 = q * (64 / E);                               // sixty-fourths of a turn, 0..31
    const cf A = a, B = b;
    if (q64 == 0) {
        a = cadd(A, B); b = csub(A, B);
        return;
    }
    if (q64 == 16) {
        const cf wb = INV ? cf{-B.y, B.x} : cf{B.y, -B.x};
        a = cadd(A, wb); b = csub(A, wb);
        return;
    }
    const float c = q64 <= 16 ? kCos64[q64] : -kCos64[32 - q64];
    const float s0 = q64 <= 16 ? kCos64[16 - q64] : kCos64[q64 - 16];
    const float sn = INV ? s0 : -s0;
    float p, g, f;
    if (c >= s0 || -c >= s0) {
        const float t = sn / c;
        p = __builtin_fmaf(-t, B.y, B.x);
        g = __builtin_fmaf(t, B.x, B.y);
        f = c;
    } else {
        const float r = c / sn;
        p = __builtin_fmaf(r, B.x, -B.y);
        g = __builtin_fmaf(r, B.y, B.x);
        f = sn;
    }
    a = cf{__builtin_fmaf(f, p, A.x), __builtin_fmaf(f, g, A.y)};
    b = cf{__builtin_fmaf(-f, p, A.x), __builtin_fmaf(-f, g, A.y)};
}

// natural order in, v[r] = X[brevE(r)] out.  The decimation-in-time network works on w[i] = v[brevE(i)] and leaves
// w[j] = X[j]; with w aliased onto v through the index map both permutations cost nothing.
template <int E, bool INV>
__device__ __forceinline__ void fftE(cf (&v)[E])
{
#pragma unroll
    for (int half = 1; half <= E / 2; half <<= 1) {
        const int tstep = (E / 2) / half;
#pragma unroll
        for (int blk = 0; blk < E; blk += 2 * half) {
#pragma unroll
            for (int q = 0; q < half; q++)
                bfly_ditE<E, INV>(v[brevE<E>(blk + q)], v[brevE<E>(blk + q + half)], q * tstep);
        }
    }
}

// the E x E core shared by the square-size kernels: E-point transform, transpose inside the lane group, inter-pass twiddle
// W_(E*E)^(k1 * l) read as contiguous rows of the symmetric table, E-point transform.  In: v[j] = element lg + E j of the
// group's transform; out: v[q] = bin lg + E brevE(q).
template <int E, bool INV>
__device__ __forceinline__ void square_core(cf (&v)[E], float *buf, const float2 *__restrict__ tw2d, int lg)
{
    constexpr int PITCH = E + 1;
    fftE<E, INV>(v);
#pragma unroll
    for (int q = 0; q < E; q++) buf[brevE<E>(q) * PITCH + lg] = v[q].x;
    OLS_WAVE_SYNC();
#pragma unroll
    for (int cidx = 0; cidx < E; cidx++) v[cidx].x = buf[lg * PITCH + cidx];
    OLS_WAVE_SYNC();
#pragma unroll
    for (int q = 0; q < E; q++) buf[brevE<E>(q) * PITCH + lg] = v[q].y;
    OLS_WAVE_SYNC();
#pragma unroll
    for (int cidx = 0; cidx < E; cidx++) v[cidx].y = buf[lg * PITCH + cidx];
    OLS_WAVE_SYNC();
#pragma unroll
    for (int l0 = 0; l0 < E; l0 += 8) {
#pragma unroll
        for (int l = l0; l < l0 + 8; l++) {
            const float2 w = tw2d[l * E + lg];
            v[l] = cmul<INV>(v[l], cf{w.x, w.y});
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    fftE<E, INV>(v);
}

// tw2d: [E][E] float2, entry [k1][l] = exp(-2 pi j k1 l / N); one transform per group of E lanes, 256 / E per workgroup
// E = 64 needs ~300 VGPRs: at least two waves per SIMD are asked for (256 registers, a few spilled), which beats one
// wave with everything in registers: N = 4096 0.248 -> 0.197 ms
template <int E, bool INV>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(E == 64 ? 2 : 1)))
k_fft_square_f32(float *__restrict__ data, int count, const float2 *__restrict__ tw2d)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int N = E * E, GROUPS = 256 / E, PITCH = E + 1;
    __shared__ float bufs[GROUPS][E * PITCH];
    const int tid = threadIdx.x, grp = tid / E, lg = tid % E;
    long t = (long)blockIdx.x * GROUPS + grp;
    if (t >= count) return;                                     // whole groups leave together: no barrier below
    // a whole wave per transform: tell the compiler the base is wave-uniform, so that the 2 x 64 accesses are scalar base +
    // one lane offset + immediates instead of 128 separate 64-bit address pairs (400+ -> ~300 VGPRs; still one wave per
    // SIMD: the 64-element network itself holds ~220, fencing the butterflies in groups did not lower it)
    if (E == 64) t = (long)blockIdx.x * GROUPS + __builtin_amdgcn_readfirstlane(grp);
    f32x2 *g = reinterpret_cast<f32x2 *>(data) + t * N;
    float *buf = bufs[grp];
    cf v[E];
#pragma unroll
    for (int j = 0; j < E; j++) {
        const f32x2 x = __builtin_nontemporal_load(&g[lg + E * j]);
        v[j] = INV ? cf{x.x * (1.0f / N), x.y * (1.0f / N)} : cf{x.x, x.y};
    }
    square_core<E, INV>(v, buf, tw2d, lg);                      // v[q] = X[lg + E brevE(q)]
#pragma unroll
    for (int q = 0; q < E; q++)
        __builtin_nontemporal_store((f32x2){v[q].x, v[q].y}, &g[lg + E * brevE<E>(q)]);
}

// N = 2 E^2 (E = 16: 512, E = 32: 2048): one radix-2 step around two E x E transforms held by the same lane group.
//   forward: s = x[n] + x[n + N/2], d = (x[n] - x[n + N/2]) W_N^n;  X[2k] = F(s)[k], X[2k+1] = F(d)[k]  (stored as one
//            16-byte pair per lane);
//   inverse: the mirror image, 1/N folded into the loads.
// tw1: [E][E] float2, entry [j][l] = W_N^(l + E j).
template <int E, bool INV>
__global__ void __launch_bounds__(256)
k_fft_2xsquare_f32(float *__restrict__ data, int count, const float2 *__restrict__ tw2d, const float2 *__restrict__ tw1)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int H = E * E, N = 2 * H, GROUPS = 256 / E, PITCH = E + 1;
    __shared__ float bufs[GROUPS][E * PITCH];
    const int tid = threadIdx.x, grp = tid / E, lg = tid % E;
    const long t = (long)blockIdx.x * GROUPS + grp;
    if (t >= count) return;
    float *base = data + t * (2 * N);
    f32x2 *g2 = reinterpret_cast<f32x2 *>(base);
    f32x4 *g4 = reinterpret_cast<f32x4 *>(base);
    float *buf = bufs[grp];
    cf s[E], d[E];
    if (!INV) {
#pragma unroll
        for (int j = 0; j < E; j++) {
            const f32x2 a = __builtin_nontemporal_load(&g2[lg + E * j]);
            const f32x2 b = __builtin_nontemporal_load(&g2[lg + E * j + H]);
            const float2 w = tw1[j * E + lg];
            s[j] = cf{a.x + b.x, a.y + b.y};
            d[j] = cmul<false>(cf{a.x - b.x, a.y - b.y}, cf{w.x, w.y});
        }
        square_core<E, false>(s, buf, tw2d, lg);
        square_core<E, false>(d, buf, tw2d, lg);
#pragma unroll
        for (int q = 0; q < E; q++)                                   // bins 2k and 2k+1, k = lg + E brevE(q)
            __builtin_nontemporal_store((f32x4){s[q].x, s[q].y, d[q].x, d[q].y}, &g4[lg + E * brevE<E>(q)]);
    } else {
        constexpr float sc = 1.0f / N;
#pragma unroll
        for (int j = 0; j < E; j++) {
            const f32x4 x = __builtin_nontemporal_load(&g4[lg + E * j]);
            s[j] = cf{x.x * sc, x.y * sc};
            d[j] = cf{x.z * sc, x.w * sc};
        }
        square_core<E, true>(s, buf, tw2d, lg);
        square_core<E, true>(d, buf, tw2d, lg);
#pragma unroll
        for (int q = 0; q < E; q++) {                                 // n = lg + E brevE(q)
            const float2 w = tw1[brevE<E>(q) * E + lg];
            const cf wd = cmul<true>(d[q], cf{w.x, w.y});             // d * conj(W_N^n)
            __builtin_nontemporal_store((f32x2){s[q].x + wd.x, s[q].y + wd.y}, &g2[lg + E * brevE<E>(q)]);
            __builtin_nontemporal_store((f32x2){s[q].x - wd.x, s[q].y - wd.y}, &g2[lg + E * brevE<E>(q) + H]);
        }
    }
}

// tables of the 2 E^2 kernel from the caller's table cs of size N = 2 E^2: tw2d[k1][l] = W_(E^2)^(k1 l) = W_N^(2 k1 l),
// tw1[j][l] = W_N^(l + E j)
__global__ void k_fft_2xsquare_tables(float2 *__restrict__ tw2d, float2 *__restrict__ tw1, const float *__restrict__ cs,
                                      int E)
{
    const int i = blockIdx.x * 256 + threadIdx.x, H = E * E, N = 2 * H;
    if (i >= H) return;
    const int m2 = (2 * (i / E) * (i % E)) & (N - 1);
    tw2d[i] = make_float2(cs[m2], -cs[N + m2]);
    const int m1 = (i % E) + E * (i / E);
    tw1[i] = make_float2(cs[m1], -cs[N + m1]);
}

// Q15 transforms (llz_fft_fixed.c:61-218) of N = E^2 (E = 8, 16, 32, 64) or 2 E^2 (TWO; E = 8, 16, 32) points on a group
// of E lanes, the layout of the float32 register kernels: the radix-2 stages run as two register passes of log2 E stages
// with one LDS transpose between them (TWO: one more stage across the two halves, which the same lanes hold) -- every
// butterfly is the reference's butterfly on the reference's operands (arith_q15::rot: four separately floored
// (int64 * int64) >> 15 products, wrapping adds), so the result is bit-identical; what goes is the index arithmetic and
// the LDS round trips of the staged passes, which is what those were bound by (0.93e12 butterflies/s at every size: ~30
// vector instructions per butterfly, 12 of them arithmetic).
// Forward (DIF), one half: element i = l + E j in register j of lane l; stages with half-span E^2/2 .. E pair registers;
// transpose; lane m then holds i = E m + c and stages E/2 .. 1 pair registers again; the bit-reversed gather becomes the
// store pattern.  Inverse (DIT): the mirror image, >> log2 N at the end (llz_fft_fixed.c:212-215).
template <int E, bool TWO, bool INV>
__global__ void __launch_bounds__(256)
k_fft_reg_q15(int *__restrict__ data, int count, const short *__restrict__ cs /* N cos, then N sin, Q15 */)
{
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    constexpr int H = E * E, N = TWO ? 2 * H : H, GROUPS = 256 / E, PITCH = E + 1, NH = TWO ? 2 : 1;
    constexpr int LOG = E == 8 ? 3 : E == 16 ? 4 : E == 32 ? 5 : 6, LN = 2 * LOG + (TWO ? 1 : 0);
    __shared__ int s_tw[N / 2];                                    // (cos, sin) of 2 pi e / N as a pair of shorts
    __shared__ int bufs[GROUPS][E * PITCH];
    const int tid = threadIdx.x, grp = tid / E, lg = tid % E;
    for (int e = tid; e < N / 2; e += 256)
        s_tw[e] = (int)((unsigned)(unsigned short)cs[e] | ((unsigned)(unsigned short)cs[N + e] << 16));
    __syncthreads();
    const long t = (long)blockIdx.x * GROUPS + grp;
    if (t >= count) return;                                        // whole groups leave together: no barrier below
    i32x2 *g = reinterpret_cast<i32x2 *>(data) + t * N;
    int *buf = bufs[grp];
    const int lrev = (int)(__brev((unsigned)lg) >> (32 - LOG));
    int vr[NH][E], vi[NH][E];
    auto transpose = [&](int h) {                                  // (lane a, register b) -> (lane b, register a)
#pragma unroll
        for (int b = 0; b < E; b++) buf[b * PITCH + lg] = vr[h][b];
        OLS_WAVE_SYNC();
#pragma unroll
        for (int b = 0; b < E; b++) vr[h][b] = buf[lg * PITCH + b];
        OLS_WAVE_SYNC();
#pragma unroll
        for (int b = 0; b < E; b++) buf[b * PITCH + lg] = vi[h][b];
        OLS_WAVE_SYNC();
#pragma unroll
        for (int b = 0; b < E; b++) vi[h][b] = buf[lg * PITCH + b];
        OLS_WAVE_SYNC();
    };
    // one butterfly of the reference on (ar, ai), (br, bi) with the table entry idx
    auto bfly = [&](int &ar, int &ai, int &br, int &bi, int idx) {
        const int tw = s_tw[idx];
        const short wr = (short)(tw & 0xffff), ws = (short)(tw >> 16);
        const int ur = ar, ui = ai, wre = br, wim = bi;
        if (!INV) {                                                // llz_fft_fixed.c:76-92
            int yr, yi;
            arith_q15::rot(arith_q15::sub(ur, wre), arith_q15::sub(ui, wim), wr, arith_q15::neg(ws), yr, yi);
            ar = arith_q15::add(ur, wre); ai = arith_q15::add(ui, wim);
            br = yr; bi = yi;
        } else {                                                   // llz_fft_fixed.c:122-137
            int dr, di;
            arith_q15::rot(wre, wim, wr, ws, dr, di);
            ar = arith_q15::add(ur, dr); ai = arith_q15::add(ui, di);
            br = arith_q15::sub(ur, dr); bi = arith_q15::sub(ui, di);
        }
    };
    if (!INV) {
#pragma unroll
        for (int h = 0; h < NH; h++)
#pragma unroll
            for (int j = 0; j < E; j++) { const i32x2 x = g[lg + E * j + H * h]; vr[h][j] = x.x; vi[h][j] = x.y; }
        if (TWO) {                                                 // half-span N/2: the two halves of a lane
#pragma unroll
            for (int j = 0; j < E; j++) bfly(vr[0][j], vi[0][j], vr[NH - 1][j], vi[NH - 1][j], lg + E * j);
        }
#pragma unroll
        for (int h = 0; h < NH; h++) {
#pragma unroll
            for (int gq = 0; gq < LOG; gq++) {                     // half-spans E^2/2 .. E: registers hj apart
                const int hj = (E / 2) >> gq;
#pragma unroll
                for (int j = 0; j < E; j++)
                    if (!(j & hj)) bfly(vr[h][j], vi[h][j], vr[h][j + hj], vi[h][j + hj],
                                        (lg + E * (j & (hj - 1))) << (gq + (TWO ? 1 : 0)));
            }
            transpose(h);                                          // lane m: register c = element E m + c of the half
#pragma unroll
            for (int gq = 0; gq < LOG; gq++) {                     // half-spans E/2 .. 1
                const int hc = (E / 2) >> gq;
#pragma unroll
                for (int c = 0; c < E; c++)
                    if (!(c & hc)) bfly(vr[h][c], vi[h][c], vr[h][c + hc], vi[h][c + hc], (c & (hc - 1)) << (LN - LOG + gq));
            }
        }
#pragma unroll
        for (int h = 0; h < NH; h++)                               // llz_fft_fixed.c:170-174: out[brev(i)] = work[i]
#pragma unroll
            for (int c = 0; c < E; c++) g[((brevE<E>(c) * E + lrev) * NH) + h] = (i32x2){vr[h][c], vi[h][c]};
    } else {
#pragma unroll
        for (int h = 0; h < NH; h++)
#pragma unroll
            for (int c = 0; c < E; c++) { const i32x2 x = g[((brevE<E>(c) * E + lrev) * NH) + h]; vr[h][c] = x.x; vi[h][c] = x.y; }
#pragma unroll
        for (int h = 0; h < NH; h++) {
#pragma unroll
            for (int gq = 0; gq < LOG; gq++) {                     // half-spans 1 .. E/2
                const int hc = 1 << gq;
#pragma unroll
                for (int c = 0; c < E; c++)
                    if (!(c & hc)) bfly(vr[h][c], vi[h][c], vr[h][c + hc], vi[h][c + hc], (c & (hc - 1)) << (LN - 1 - gq));
            }
            transpose(h);                                          // lane l: register j = element l + E j of the half
#pragma unroll
            for (int gq = 0; gq < LOG; gq++) {                     // half-spans E .. E^2/2
                const int hj = 1 << gq;
#pragma unroll
                for (int j = 0; j < E; j++)
                    if (!(j & hj)) bfly(vr[h][j], vi[h][j], vr[h][j + hj], vi[h][j + hj],
                                        (lg + E * (j & (hj - 1))) << (LN - 1 - LOG - gq));
            }
        }
        if (TWO) {
#pragma unroll
            for (int j = 0; j < E; j++) bfly(vr[0][j], vi[0][j], vr[NH - 1][j], vi[NH - 1][j], lg + E * j);
        }
#pragma unroll
        for (int h = 0; h < NH; h++)
#pragma unroll
            for (int j = 0; j < E; j++) g[lg + E * j + H * h] = (i32x2){vr[h][j] >> LN, vi[h][j] >> LN};   // :212-215
    }
}

// Analysis frames for fft_len = E^2 (E = 16: 256) or 2 E^2 (TWO; E = 16: 512, E = 32: 2048) on a group of E lanes per
// frame: k_stft_analysis1024_f32's scheme on square_core -- windowed samples from HBM straight into the registers of the
// lane that transforms them, bins 0..size/2 straight back.
// (fft_len 2048 needs 273 VGPRs: two waves per SIMD asked for, 0.68 -> 0.63 ms)
template <int E, bool TWO>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((E == 32 && TWO) ? 2 : 1)))
k_stft_analysis_reg_f32(const float *__restrict__ x, const float *__restrict__ hist, float *__restrict__ re,
                        float *__restrict__ im, const float *__restrict__ w, int frames, int F,
                        const float2 *__restrict__ tw2d, const float2 *__restrict__ tw1, long x_pitch, long total_tr)
{
    constexpr int H = E * E, SIZE = TWO ? 2 * H : H, GROUPS = 256 / E, PITCH = E + 1, BINS = SIZE / 2 + 1;
    __shared__ float bufs[GROUPS][E * PITCH];
    const int tid = threadIdx.x, grp = tid / E, lg = tid % E;
    const long g = (long)blockIdx.x * GROUPS + grp;
    if (g >= total_tr) return;
    const int c = (int)(g / frames), f = (int)(g - (long)c * frames);
    const int keep = SIZE - F;
    const long t0 = (long)(f + 1) * F - SIZE;
    const float *row = x + (size_t)c * x_pitch;
    const float *hrow = hist + (size_t)c * keep;
    float *buf = bufs[grp];
    auto sample = [&](int i) {
        const long t = t0 + i;
        return (t >= 0 ? row[t] : hrow[keep + t]) * w[i];
    };
    const size_t o = (size_t)g * BINS;
    if (!TWO) {
        cf v[E];
#pragma unroll
        for (int j = 0; j < E; j++) v[j] = cf{sample(lg + E * j), 0.f};
        square_core<E, false>(v, buf, tw2d, lg);                    // v[q] = X[lg + E brevE(q)]
#pragma unroll
        for (int q = 0; q < E; q++) {
            const int bin = lg + E * brevE<E>(q);
            if (bin < BINS) { re[o + bin] = v[q].x; im[o + bin] = v[q].y; }
        }
    } else {
        cf s[E], d[E];
#pragma unroll
        for (int j = 0; j < E; j++) {                               // real input: s, d before the twist are real
            const float a = sample(lg + E * j), b = sample(lg + E * j + H);
            const float2 t = tw1[j * E + lg];
            s[j] = cf{a + b, 0.f};
            d[j] = cf{(a - b) * t.x, (a - b) * t.y};
        }
        square_core<E, false>(s, buf, tw2d, lg);                    // s[q] = X[2 kq], d[q] = X[2 kq + 1], kq = lg + E brevE(q)
        square_core<E, false>(d, buf, tw2d, lg);
#pragma unroll
        for (int q = 0; q < E; q++) {
            const int bin = 2 * (lg + E * brevE<E>(q));
            if (bin < BINS) { re[o + bin] = s[q].x; im[o + bin] = s[q].y; }
            if (bin + 1 < BINS) { re[o + bin + 1] = d[q].x; im[o + bin + 1] = d[q].y; }
        }
    }
}

// Synthesis frames for the same sizes: k_stft_synthesis1024_f32's walk (bins from HBM into registers with the Hermitian
// upper half taken from the mirrored bin, inverse transform, windowed real output into an LDS segment image, overlap-add
// in the reference's order, oldest frame first) with 256 / E frames per group on square_core.
// HALF (fft_len = 2 E^2): the spectrum of a REAL frame of N = 2H samples needs ONE H-point complex inverse transform, not two:
// with E[k] = (X[k] + conj(X[H-k])) / 2 and O[k] = (X[k] - conj(X[H-k])) W_N^-k / 2 (the spectra of the even and the odd samples),
// z = IDFT_H(E + j O) is x[2n] + j x[2n+1].  Both bins come from HBM (the mirrored one by its own index: no lane exchange), the
// results leave as (even, odd) pairs: half the transform work of the TWO form (1024 ch x 128 frames at 3/4 overlap, fft_len 2048:
// 1.89 -> 1.15 ms; fft_len 512, 256 frames at 1/2 overlap: 0.48 -> 0.33 ms).  cs: cos, then sin of 2 pi i / N.
// (At fft_len 2048 eight frame images are 64 KB and allow ONE workgroup per CU; letting the frames enter a four-frame image in
//  two parts -- 75 KB, two workgroups per CU, two more barriers per group -- measured slower, 1.15 -> 1.32 ms.)
template <int E, bool TWO, bool HALF>
__global__ void __launch_bounds__(256)
k_stft_synthesis_reg_f32(const float *__restrict__ re, const float *__restrict__ im, float *__restrict__ x,
                         const float *__restrict__ ola_old, float *__restrict__ ola_new, const float *__restrict__ w,
                         int frames, int F, const float2 *__restrict__ tw2d, const float2 *__restrict__ tw1, long x_pitch,
                         int run_len, int runs, float magic, const float *__restrict__ cs)
{
    static_assert(!(TWO && HALF), "the half-size form runs one square transform");
    constexpr int H = E * E, N = (TWO || HALF) ? 2 * H : H, TPW = 256 / E, PITCH = E + 1, BINS = N / 2 + 1;
    constexpr int MAXM = ((TPW - 1) * (N / 2) + N + 255) / 256;       // span of a group at the largest hop (N/2)
    __shared__ float bufs[TPW][E * PITCH];
    __shared__ float seg[TPW][N];
    __shared__ float carry[N];                                         // N - F used
    const int tid = threadIdx.x, grp = tid / E, lg = tid % E;
    const int c = blockIdx.x / runs, run = blockIdx.x - c * runs;
    const int b0 = run * run_len, b1 = min(frames, b0 + run_len);
    const int R = N / F, keep = N - F;
    const int fs = max(0, b0 - (R - 1));
    for (int q = tid; q < keep; q += 256) carry[q] = fs == 0 ? ola_old[(size_t)c * keep + q] : 0.f;
    constexpr float sc = 1.0f / (float)N;
    for (int g0 = fs; g0 < b1; g0 += TPW) {
        const int ng = min(TPW, b1 - g0);
        __syncthreads();                                               // carry and seg of the previous group are consumed
        if (grp < ng) {
            const size_t o = ((size_t)c * frames + g0 + grp) * BINS;
            auto bin = [&](int k) {                                    // upper half: conjugate of the mirrored bin
                const int kk = k <= N / 2 ? k : N - k;
                const float a = re[o + kk] * sc, b = im[o + kk] * sc;
                return cf{a, k <= N / 2 ? b : -b};
            };
            float *buf = bufs[grp];
            if constexpr (HALF) {
                cf v[E];
#pragma unroll
                for (int j = 0; j < E; j++) {
                    const int k = lg + E * j;                          // 0 .. H-1; its partner H - k is in 1 .. H
                    // (bins 0 and H are real in the spectrum of a real frame; whatever their imaginary parts hold reaches only the
                    //  imaginary output of the full-size transform, which is dropped: the same here)
                    const float xr = re[o + k], xi = k == 0 ? 0.f : im[o + k], mr = re[o + H - k], mi = k == 0 ? 0.f : -im[o + H - k];
                    const float sr = xr + mr, si = xi + mi, dr = xr - mr, di = xi - mi;
                    const float cw = cs[k], sn = cs[N + k];            // W_N^-k = cw + j sn
                    const float orr = dr * cw - di * sn, oi = dr * sn + di * cw;     // (X[k] - conj X[H-k]) W_N^-k
                    v[j] = cf{(sr - oi) * sc, (si + orr) * sc};        // (E + j O) / H = (S + j D W) / N
                }
                square_core<E, true>(v, buf, tw2d, lg);                // v[q] = x[2n] + j x[2n+1], n = lg + E brevE(q)
#pragma unroll
                for (int q = 0; q < E; q++) {
                    const int n = lg + E * brevE<E>(q);
                    const float2 ww = *reinterpret_cast<const float2 *>(w + 2 * n);
                    *reinterpret_cast<float2 *>(&seg[grp][2 * n]) = make_float2(v[q].x * ww.x, v[q].y * ww.y);
                }
            } else if (!TWO) {
                cf v[E];
#pragma unroll
                for (int j = 0; j < E; j++) v[j] = bin(lg + E * j);
                square_core<E, true>(v, buf, tw2d, lg);                // v[q].x = sample lg + E brevE(q)
#pragma unroll
                for (int q = 0; q < E; q++) {
                    const int n = lg + E * brevE<E>(q);
                    seg[grp][n] = v[q].x * w[n];
                }
            } else {
                cf s[E], d[E];
#pragma unroll
                for (int j = 0; j < E; j++) {
                    s[j] = bin(2 * (lg + E * j));
                    d[j] = bin(2 * (lg + E * j) + 1);
                }
                square_core<E, true>(s, buf, tw2d, lg);
                square_core<E, true>(d, buf, tw2d, lg);
#pragma unroll
                for (int q = 0; q < E; q++) {                          // n = lg + E brevE(q): x[n], x[n + H] = s +- d conj(W_N^n)
                    const int n = lg + E * brevE<E>(q);
                    const float2 t = tw1[brevE<E>(q) * E + lg];
                    const float wdx = __builtin_fmaf(d[q].y, t.y, d[q].x * t.x);   // Re(d * conj(t))
                    seg[grp][n] = (s[q].x + wdx) * w[n];
                    seg[grp][n + H] = (s[q].x - wdx) * w[n + H];
                }
            }
        }
        __syncthreads();
        const int span = (ng - 1) * F + N;
        float acc[MAXM];
#pragma unroll
        for (int m = 0; m < MAXM; m++) {
            const int p = tid + m * 256;
            float a = 0.f;
            if (p < span) {
                a = p < keep ? carry[p] : 0.f;
                const int k_hi = min(ng - 1, p / F);
                const int k_lo = p < N ? 0 : (p - N) / F + 1;
                for (int k = k_lo; k <= k_hi; k++) a += seg[k][p - k * F];
            }
            acc[m] = a;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MAXM; m++) {
            const int p = tid + m * 256;
            if (p < span) {
                if (p < ng * F) {
                    if (g0 + p / F >= b0) x[(size_t)c * x_pitch + (size_t)g0 * F + p] = magic * acc[m];
                } else {
                    carry[p - ng * F] = acc[m];
                }
            }
        }
    }
    __syncthreads();
    if (b1 == frames)
        for (int q = tid; q < keep; q += 256) ola_new[(size_t)c * keep + q] = carry[q];
}

template <int E>
__global__ void __launch_bounds__(256)
k_acf_sq_f32(const float *__restrict__ x, float *__restrict__ r, int frames, int n, int p,
             const float2 *__restrict__ tw2d, const float *__restrict__ cs /* 2H cos, then 2H sin of 2 pi i / (2H) */)
{
    constexpr int H = E * E, F = 2 * H, GROUPS = 256 / E, PITCH = E + 1;
    __shared__ float bufs[GROUPS][E * PITCH];
    const int tid = threadIdx.x, grp = tid / E, lg = tid % E;
    const long t = (long)blockIdx.x * GROUPS + grp;
    if (t >= frames) return;
    const float *g = x + t * n;
    float *buf = bufs[grp];
    cf v[E];
#pragma unroll
    for (int j = 0; j < E; j++) {
        const int i0 = 2 * (lg + E * j);
        v[j].x = (j < E / 2 && i0 < n) ? g[i0] : 0.f;              // 2 m < H: the upper half of z is padding
        v[j].y = (j < E / 2 && i0 + 1 < n) ? g[i0 + 1] : 0.f;
    }
    square_core<E, false>(v, buf, tw2d, lg);                       // v[q] = Z[lg + E brevE(q)]
    const int lm = (E - lg) % E;
    float mx[E];
#pragma unroll
    for (int q = 0; q < E; q++) buf[brevE<E>(q) * PITCH + lg] = v[q].x;
    OLS_WAVE_SYNC();
#pragma unroll
    for (int q = 0; q < E; q++) {
        const int j = brevE<E>(q);
        mx[q] = buf[lg ? (E - 1 - j) * PITCH + lm : ((E - j) % E) * PITCH];
    }
    OLS_WAVE_SYNC();
#pragma unroll
    for (int q = 0; q < E; q++) buf[brevE<E>(q) * PITCH + lg] = v[q].y;
    OLS_WAVE_SYNC();
    const float sc = 1.0f / (4.0f * (float)F);
#pragma unroll
    for (int q = 0; q < E; q++) {
        const int j = brevE<E>(q);
        const int k = lg + E * j;
        const float my = buf[lg ? (E - 1 - j) * PITCH + lm : ((E - j) % E) * PITCH];
        const float wc = cs[k], ws = -cs[F + k];                   // W_F^k = (cos, -sin)
        const cf xe = {v[q].x + mx[q], v[q].y - my};
        const cf xo = {v[q].y + my, mx[q] - v[q].x};
        const cf T = cmul<false>(xo, cf{wc, ws});
        const cf a = cadd(xe, T), b = csub(xe, T);
        float sk = __builtin_fmaf(a.x, a.x, a.y * a.y) * sc, sm = __builtin_fmaf(b.x, b.x, b.y * b.y) * sc;
        if (k >= n) sk = 0.f;
        if (H - k >= n) sm = 0.f;
        if (k == 0) { sk *= 2.f; sm = 0.f; }
        const float dk = sk - sm;
        v[q] = cf{__builtin_fmaf(ws, dk, sk + sm), wc * dk};
    }
    OLS_WAVE_SYNC();
    cf u[E];
#pragma unroll
    for (int j = 0; j < E; j++) u[j] = v[brevE<E>(j)];
    square_core<E, true>(u, buf, tw2d, lg);                        // u[q] = g[lg + E brevE(q)]
    float *rr = r + t * (p + 1);
#pragma unroll
    for (int q = 0; q < E; q++) {
        const int m = lg + E * brevE<E>(q);
        if (2 * m <= p) rr[2 * m] = u[q].x;
        if (2 * m + 1 <= p) rr[2 * m + 1] = u[q].y;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// MDCT by the N/4-point FFT (llz_mdct.c:266-353, mdct2 / imdct2) on the register transforms: a group of E lanes per
// frame, N/4 = E^2 (TWO = false: N = 256, 1024, 4096) or 2 E^2 (TWO = true: N = 512, 2048, 8192).  The pre-twiddled
// points are formed straight from HBM into the registers of the lane that transforms them and the post-twiddled results
// go straight back (the rotations 2k / N-1-2k are stride-2 walks up and down the same rows: every line is used whole,
// half by each walk), so the only LDS traffic left is the transpose inside the group.
//   forward: in = x [count][N], out = X [count][N/2];   inverse: in = X [count][N/2], out = x [count][N]
// tc/ts: cos and sin of -2 pi (k + 1/8) / N, k < N/4 (llz_mdct.c:459-462).  Both directions use the FORWARD transform.
// (N = 8192 forward needs 300 VGPRs: two waves per SIMD are forced there, the little that does not fit is spilled --
//  0.94 -> 0.51 ms; the inverse at 344 loses with the same treatment, 0.53 -> 0.57 ms, and keeps one wave)
// FRAMES: the windowed 50 %-overlap frames of llz_asmodel.c:313-463 in batch form (time-domain alias cancellation).  A block
// is (channel c, frame f) of a planar signal [channels][frames F], F = N/2:
//   forward: the transform's input is w[i] xbuf[i], xbuf = the previous frame followed by frame f = the signal from
//            (f-1) F on (frame 0: its first half is the handle's state, the last frame of the previous call); the block of the
//            last frame leaves that frame in state_out;
//   inverse: w[j] y[j] is ADDED to the signal from f F on (llz_asmodel.c:451-452).  Two launches: the even frames store
//            (their ranges [f F, (f+2) F) tile the signal; frame 0 adds the previous call's tail), then the odd frames add to
//            what is there; the last frame's second half is the new tail (state_out).  A sum of two terms does not depend on
//            their order, so the result is that of the reference's sequential overlap-add.
struct mdct_fr {
    const float *win;          // [N]
    const float *state_in;     // [channels][F]
    float *state_out;          // [channels][F], not the same buffer
    int frames;                // frames per channel in this call
    int first, step;           // this launch: frames first, first + step, ...
    int per_channel;           // how many of them (FR = 2: how many runs) per channel
    int run;                   // FR = 2: output segments per group
};

// FR: 0 plain batch, 1 frames (a group per frame), 2 frames, inverse only: a group per RUN of consecutive segments
template <int E, bool TWO, bool INVERSE, int FR>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((E == 32 && TWO && !INVERSE) ? 2 : 1)))
k_mdct_reg_f32(const float *__restrict__ in, float *__restrict__ out, int count, const float *__restrict__ tc,
               const float *__restrict__ ts, const float2 *__restrict__ tw2d, const float2 *__restrict__ tw1,
               float sqrt_cof, mdct_fr fr)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int H = E * E, N4 = TWO ? 2 * H : H, N = 4 * N4, N2 = N / 2, GROUPS = 256 / E, PITCH = E + 1;
    constexpr bool FRAMES = FR != 0, RUN = FR == 2;
    constexpr int VW = TWO ? 4 : 2;                                 // floats per lane and store of the inverse's output
    static_assert(!RUN || INVERSE, "runs of segments are a synthesis form");
    __shared__ float bufs[GROUPS][E * PITCH];
    // FRAMES: the window in LDS, one copy for the workgroup's frames (through the vector memory path it was one load per
    // signal load)
    __shared__ __attribute__((aligned(16))) float s_win[FRAMES ? N : 4];
    const int tid = threadIdx.x, grp = tid / E, lg = tid % E;
    if constexpr (FRAMES) {
        for (int i = 4 * tid; i < N; i += 4 * 256) *reinterpret_cast<f32x4 *>(s_win + i) = *reinterpret_cast<const f32x4 *>(fr.win + i);
        __syncthreads();
    }
    const long t = (long)blockIdx.x * GROUPS + grp;
    if (t >= count) return;                                        // whole groups leave together: no barrier below
    const float *x;
    float *y;
    int fc = 0, ff = 0;                                            // FRAMES: channel and frame of this block
    int s0 = 0, f_end = 0;                                         // RUN: segments s0 .. f_end - 1 are this group's
    float tail[RUN ? N2 / E : 1];                                  // RUN: the previous frame's windowed second half, in the
                                                                   // lane's own store layout (VW floats per VW E outputs)
    if constexpr (RUN) {
        fc = (int)(t / fr.per_channel);
        s0 = fr.run * (int)(t - (long)fc * fr.per_channel);
        f_end = min(s0 + fr.run, fr.frames);
        ff = s0 ? s0 - 1 : 0;                                      // the frame in front is transformed again for its tail
        const float *prev = fr.state_in + (size_t)fc * N2;         // segment 0 starts from the previous call's tail
#pragma unroll
        for (int i = 0; i < N2 / E; i += VW) {
#pragma unroll
            for (int c = 0; c < VW; c++) tail[i + c] = s0 ? 0.f : prev[i * E + VW * lg + c];
        }
    }
    for (;;) {                                                      // (one trip unless RUN)
    if constexpr (FRAMES) {
        if constexpr (!RUN) {
            fc = (int)(t / fr.per_channel);
            ff = fr.first + fr.step * (int)(t - (long)fc * fr.per_channel);
        }
        const size_t sig = (size_t)fc * fr.frames * N2;            // the channel's signal: frames * F samples, F = N2
        if (!INVERSE) {
            x = in + sig + (long)(ff - 1) * N2;                    // xbuf[i] = x[i] (frame 0: i < F comes from the state)
            y = out + (sig + (size_t)ff * N2);                     // coefficients [channels][frames][F]
        } else {
            x = in + (sig + (size_t)ff * N2);
            y = out + sig + (size_t)ff * N2;                       // y[j], j < 2F: the signal from f F on
        }
    } else {
        x = in + t * (INVERSE ? N2 : N);
        y = out + t * (INVERSE ? N : N2);
    }
    const float *st_in = FRAMES ? fr.state_in + (size_t)fc * N2 : nullptr;
    float *st_out = FRAMES ? fr.state_out + (size_t)fc * N2 : nullptr;
    const bool last = FRAMES && ff == fr.frames - 1;
    float *buf = bufs[grp];
    // The point k and its mirror N/4-1-k share their rows pairwise: x[2k] goes to k, x[2k+1] to the mirror, and so on.
    // The mirror of (lane lg, register j) is (lane E-1-lg, register E-1-j) -- in bin order (lane E-1-lg, register q^(E-1))
    // -- so every HBM access is an aligned pair (or quad) per lane and the other half changes lanes by one swizzle.
    auto mir = [](float v) {
        return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), ((E - 1) << 10) | 0x1F));   // lane ^ (E-1)
    };
    // rot[i] = -x[i + 3N/4] (i < N/4), x[i - N/4] otherwise (llz_mdct.c:279-283); pairs never straddle N/4
    // (ib = the part of i that does not depend on the lane, a multiple of 2E: which quarter i falls in is decided on it, at
    //  compile time after unrolling -- every access is then base + lane offset + constant)
    const float *xlo = x;                                          // FRAMES forward: where xbuf[0 .. F) lives
    if constexpr (FRAMES && !INVERSE) xlo = ff == 0 ? st_in : x;
    auto rot2 = [&](int ib, int lane2) {
        const bool neg = ib < N4;
        const int idx = (neg ? ib + 3 * N4 : ib - N4) + lane2;
        const bool hi_half = neg || ib >= 3 * N4;                  // idx >= N/2
        f32x2 v;
        if constexpr (FRAMES && !INVERSE) {
            v = *reinterpret_cast<const f32x2 *>((hi_half ? x : xlo) + idx);
            if (hi_half && last) *reinterpret_cast<f32x2 *>(st_out + idx - N2) = v;       // the next call's previous frame
            v *= *reinterpret_cast<const f32x2 *>(s_win + idx);
        } else {
            v = *reinterpret_cast<const f32x2 *>(x + idx);
        }
        return neg ? -v : v;
    };
    // z[k] = 0.5 (re + j im) (c + j s), (c, s) = cos, sin of -2 pi (k + 1/8) / N
    auto pre = [&](int k, float re, float im) {
        const float c = tc[k], sn = ts[k];
        return cf{0.5f * (re * c - im * sn), 0.5f * (re * sn + im * c)};
    };
    cf s[E], d[E];                                                 // d: the odd-bin half of TWO (unused otherwise)
#pragma unroll
    for (int j = 0; j < E / 2; j++) {
        const int jm = E - 1 - j;
        // square: points k = lg + E j and lg + E jm.  TWO: a-points n = lg + E j (registers of s) and b-points n + H (d);
        // the mirror of an a-point is a b-point and vice versa.
        const int ka = lg + E * j, kb = lg + E * jm;
        if (!INVERSE) {
            // own: r0 = rot[2k], r3 = rot[N/2+2k];  from the mirror: r2 = rot[2k+1] (= rot[N/2-1-2k']), r1 = rot[N/2+2k+1]
            if (!TWO) {
                const f32x2 Aa = rot2(2 * E * j, 2 * lg), Ba = rot2(N2 + 2 * E * j, 2 * lg), Ab = rot2(2 * E * jm, 2 * lg), Bb = rot2(N2 + 2 * E * jm, 2 * lg);
                const float r2a = mir(Ab.y), r1a = mir(Bb.y), r2b = mir(Aa.y), r1b = mir(Ba.y);
                s[j] = pre(ka, Aa.x - r1a, r2a - Ba.x);
                s[jm] = pre(kb, Ab.x - r1b, r2b - Bb.x);
            } else {
                const f32x2 Aa = rot2(2 * E * j, 2 * lg), Ba = rot2(N2 + 2 * E * j, 2 * lg), Ab = rot2(2 * E * jm, 2 * lg), Bb = rot2(N2 + 2 * E * jm, 2 * lg);
                const f32x2 Ca = rot2(2 * E * j + 2 * H, 2 * lg), Da = rot2(N2 + 2 * E * j + 2 * H, 2 * lg);
                const f32x2 Cb = rot2(2 * E * jm + 2 * H, 2 * lg), Db = rot2(N2 + 2 * E * jm + 2 * H, 2 * lg);
                // a-point ka <- mirror lane's b-point kb + H (its register jm = our j's partner): C/D of index b
                const cf a0 = pre(ka, Aa.x - mir(Db.y), mir(Cb.y) - Ba.x);
                const cf b0 = pre(ka + H, Ca.x - mir(Bb.y), mir(Ab.y) - Da.x);
                const cf a1 = pre(kb, Ab.x - mir(Da.y), mir(Ca.y) - Bb.x);
                const cf b1 = pre(kb + H, Cb.x - mir(Ba.y), mir(Aa.y) - Db.x);
                const float2 w0 = tw1[j * E + lg], w1 = tw1[jm * E + lg];
                s[j] = cadd(a0, b0); d[j] = cmul<false>(csub(a0, b0), cf{w0.x, w0.y});
                s[jm] = cadd(a1, b1); d[jm] = cmul<false>(csub(a1, b1), cf{w1.x, w1.y});
            }
        } else {
            // re = X[2k] (own), im = X[N/2-1-2k] = X[2k'+1] of the mirror (llz_mdct.c:322-324)
            if (!TWO) {
                const f32x2 Pa = *reinterpret_cast<const f32x2 *>(x + 2 * ka), Pb = *reinterpret_cast<const f32x2 *>(x + 2 * kb);
                s[j] = pre(ka, Pa.x, mir(Pb.y));
                s[jm] = pre(kb, Pb.x, mir(Pa.y));
            } else {
                const f32x2 Pa = *reinterpret_cast<const f32x2 *>(x + 2 * ka), Pb = *reinterpret_cast<const f32x2 *>(x + 2 * kb);
                const f32x2 Qa = *reinterpret_cast<const f32x2 *>(x + 2 * (ka + H)), Qb = *reinterpret_cast<const f32x2 *>(x + 2 * (kb + H));
                const cf a0 = pre(ka, Pa.x, mir(Qb.y)), b0 = pre(ka + H, Qa.x, mir(Pb.y));
                const cf a1 = pre(kb, Pb.x, mir(Qa.y)), b1 = pre(kb + H, Qb.x, mir(Pa.y));
                const float2 w0 = tw1[j * E + lg], w1 = tw1[jm * E + lg];
                s[j] = cadd(a0, b0); d[j] = cmul<false>(csub(a0, b0), cf{w0.x, w0.y});
                s[jm] = cadd(a1, b1); d[jm] = cmul<false>(csub(a1, b1), cf{w1.x, w1.y});
            }
        }
    }
    square_core<E, false>(s, buf, tw2d, lg);                       // square: s[q] = Z[lg + E brevE(q)]
    if constexpr (TWO) square_core<E, false>(d, buf, tw2d, lg);    // TWO: s[q] = Z[2 kq], d[q] = Z[2 kq + 1], kq = lg + E brevE(q)
    // post-twiddle: (c + j s) v
    auto post = [&](int k, cf v) {
        const float c = tc[k], sn = ts[k];
        return cf{v.x * c - v.y * sn, v.x * sn + v.y * c};
    };
    // rot[ri], rot[ri+1] = (v0, v1) -> x (llz_mdct.c:331-352): x[i] = rot[N/4 + i] cof (i < 3N/4), -rot[i - 3N/4] cof else
    // FRAMES: windowed, then stored / added into the signal or left as the new overlap-add tail (see mdct_fr)
    // (jb / rb = the lane-independent part of the index, a multiple of 2E: which half / quarter it falls in is decided on it
    //  at compile time, as in rot2)
    float *ylo = y;                                                 // FRAMES inverse: second half of the last frame -> tail
    auto put = [&](int jb, int lane_off, auto v) {
        typedef decltype(v) vec;
        const int j = jb + lane_off;
        if constexpr (RUN) {
            // first halves are finished with the tail of the frame before; second halves become the tail.  (The caller puts
            // j < N/2 before j + N/2: they share the tail's slot.)
            constexpr int W = sizeof(vec) / sizeof(float);
            v *= *reinterpret_cast<const vec *>(s_win + j);
            const int slot = (jb & (N2 - 1)) / E;                    // = (jb / (W E)) W: a constant after unrolling
            if (jb >= N2) {
                if (last) *reinterpret_cast<vec *>(st_out + j - N2) = v;
#pragma unroll
                for (int c = 0; c < W; c++) tail[slot + c] = v[c];
            } else if (ff >= s0) {
#pragma unroll
                for (int c = 0; c < W; c++) v[c] += tail[slot + c];
                *reinterpret_cast<vec *>(y + j) = v;
            }
            return;
        } else if constexpr (FRAMES && INVERSE) {
            v *= *reinterpret_cast<const vec *>(s_win + j);
            if (jb >= N2) {
                if (last) {
                    *reinterpret_cast<vec *>(st_out + j - N2) = v;
                    return;
                }
                if (fr.first) v += *reinterpret_cast<const vec *>(y + j);                // odd frames: add to the even frames' stores
            } else {
                if (fr.first) v += *reinterpret_cast<const vec *>(y + j);
                else if (ff == 0) v += *reinterpret_cast<const vec *>(st_in + j);         // the previous call's tail
            }
        }
        *reinterpret_cast<vec *>(ylo + j) = v;
    };
    auto unrot2 = [&](int rb, int lane_off, float v0, float v1) {
        if (rb >= N4) put(rb - N4, lane_off, (f32x2){v0 * sqrt_cof, v1 * sqrt_cof});
        else put(rb + 3 * N4, lane_off, (f32x2){-v0 * sqrt_cof, -v1 * sqrt_cof});
    };
    auto unrot4 = [&](int rb, int lane_off, float v0, float v1, float v2, float v3) {
        if (rb >= N4) put(rb - N4, lane_off, (f32x4){v0, v1, v2, v3} * sqrt_cof);
        else put(rb + 3 * N4, lane_off, (f32x4){v0, v1, v2, v3} * -sqrt_cof);
    };
    // rot[b ...] and rot[N/2 + b ...] land half a frame apart: RUN needs the one in the first half put first
    auto both2 = [&](int b, float a0, float a1, float c0, float c1) {
        if (RUN && b < N4) { unrot2(N2 + b, 2 * lg, c0, c1); unrot2(b, 2 * lg, a0, a1); }
        else { unrot2(b, 2 * lg, a0, a1); unrot2(N2 + b, 2 * lg, c0, c1); }
    };
    auto both4 = [&](int b, f32x4 a, f32x4 c) {
        if (RUN && b < N4) { unrot4(N2 + b, 4 * lg, c.x, c.y, c.z, c.w); unrot4(b, 4 * lg, a.x, a.y, a.z, a.w); }
        else { unrot4(b, 4 * lg, a.x, a.y, a.z, a.w); unrot4(N2 + b, 4 * lg, c.x, c.y, c.z, c.w); }
    };
#pragma unroll
    for (int q = 0; q < E; q++) {
        const int qm = q ^ (E - 1);
        if (q > qm) continue;                                      // pairs (q, qm): bins kq and (on the mirror lane) N4-1-kq
        const int kq = lg + E * brevE<E>(q), km = lg + E * brevE<E>(qm);
        if (!INVERSE) {
            // X[2b] = 2 Re', X[N/2-1-2b] = -2 Im' (llz_mdct.c:296-301); X[2b+1] is the -2 Im' of the mirror bin
            if (!TWO) {
                const cf pa = post(kq, s[q]), pb = post(km, s[qm]);
                const float oa = mir(-2.f * pb.y), ob = mir(-2.f * pa.y);
                *reinterpret_cast<f32x2 *>(y + 2 * kq) = (f32x2){2.f * pa.x, oa};
                *reinterpret_cast<f32x2 *>(y + 2 * km) = (f32x2){2.f * pb.x, ob};
            } else {
                const cf sa = post(2 * kq, s[q]), da = post(2 * kq + 1, d[q]), sb = post(2 * km, s[qm]), db = post(2 * km + 1, d[qm]);
                // the mirror of an even bin is an odd bin of the mirror lane's partner register, and vice versa
                const float e0 = mir(-2.f * db.y), e1 = mir(-2.f * sb.y), f0 = mir(-2.f * da.y), f1 = mir(-2.f * sa.y);
                *reinterpret_cast<f32x4 *>(y + 4 * kq) = (f32x4){2.f * sa.x, e0, 2.f * da.x, e1};
                *reinterpret_cast<f32x4 *>(y + 4 * km) = (f32x4){2.f * sb.x, f0, 2.f * db.x, f1};
            }
        } else {
            // rot[2b] = re', rot[N/2+2b] = im', rot[2b+1] = -im' of the mirror bin, rot[N/2+2b+1] = -re' of the mirror bin
            const float g = 8.f * sqrt_cof;
            if (!TWO) {
                const cf pa = post(kq, s[q]), pb = post(km, s[qm]);
                const float ia = mir(pb.y), ra = mir(pb.x), ib = mir(pa.y), rb = mir(pa.x);
                const int bq = 2 * E * brevE<E>(q), bm = 2 * E * brevE<E>(qm);    // constants after unrolling
                both2(bq, g * pa.x, -g * ia, g * pa.y, -g * ra);
                both2(bm, g * pb.x, -g * ib, g * pb.y, -g * rb);
            } else {
                const cf sa = post(2 * kq, s[q]), da = post(2 * kq + 1, d[q]), sb = post(2 * km, s[qm]), db = post(2 * km + 1, d[qm]);
                const float dbi = mir(db.y), dbr = mir(db.x), sbi = mir(sb.y), sbr = mir(sb.x);
                const float dai = mir(da.y), dar = mir(da.x), sai = mir(sa.y), sar = mir(sa.x);
                const int bq = 4 * E * brevE<E>(q), bm = 4 * E * brevE<E>(qm);
                both4(bq, (f32x4){g * sa.x, -g * dbi, g * da.x, -g * sbi}, (f32x4){g * sa.y, -g * dbr, g * da.y, -g * sbr});
                both4(bm, (f32x4){g * sb.x, -g * dai, g * db.x, -g * sai}, (f32x4){g * sb.y, -g * dar, g * db.y, -g * sar});
            }
        }
    }
    if constexpr (!RUN) break;
    else if (++ff >= f_end) break;
    }
}

// [E][E] table of the H = E^2 point transform from a table cs of the DOUBLE size F = 2H: W_H^(k1 l) = W_F^(2 k1 l)
__global__ void k_acf_sq_table(float2 *__restrict__ tw2d, const float *__restrict__ cs, int E)
{
    const int i = blockIdx.x * 256 + threadIdx.x, H = E * E, F = 2 * H;
    if (i >= H) return;
    const int m = (2 * (i / E) * (i % E)) & (F - 1);
    tw2d[i] = make_float2(cs[m], -cs[F + m]);
}

// fills tw2d from the handle's table cs (cos then sin of 2 pi i / N): exactly the host-built values
__global__ void k_fft_square_table(float2 *__restrict__ tw2d, const float *__restrict__ cs, int E)
{
    const int i = blockIdx.x * 256 + threadIdx.x, N = E * E;
    if (i >= N) return;
    const int m = ((i / E) * (i % E)) & (N - 1);
    tw2d[i] = make_float2(cs[m], -cs[N + m]);
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
    // split the log2n stages into ceil(log2n/4) passes of nearly equal depth (10 -> 4+3+3, 12 -> 4+4+4, 6 -> 3+3)
    const int passes = (log2n + 3) / 4;
    unsigned groups = 0;
    for (int p = 0, left = log2n; p < passes; p++) {
        const int G = (left + (passes - p) - 1) / (passes - p);
        groups |= (unsigned)G << (4 * p);
        left -= G;
    }
    // 2048 points per workgroup pass (256 lanes x 8): several small transforms share a workgroup
    int tpw = 2048 / size;
    if (tpw < 1) tpw = 1;
    if (tpw > count) tpw = count;
    const int tstride = size + (size >> 5) + 1;
    const size_t lds = (size_t)tpw * tstride * 2 * sizeof(typename A::data_t) +
                       (size_t)tw_entries(size) * 2 * sizeof(typename A::tw_t);
    if (lds >= 64 * 1024) {
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft_radix2<A, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fft_radix2<A, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const unsigned blocks = (unsigned)((count + tpw - 1) / tpw);
    if (inverse)
        hipLaunchKernelGGL((k_fft_radix2<A, true>), dim3(blocks), dim3(FFT_THREADS), lds, as_stream(stream), data,
                           count, size, log2n, cs, tpw, groups);
    else
        hipLaunchKernelGGL((k_fft_radix2<A, false>), dim3(blocks), dim3(FFT_THREADS), lds, as_stream(stream), data,
                           count, size, log2n, cs, tpw, groups);
    LLZ_LAUNCH_CHECK(name);
    return LLZ_OK;
}

} // namespace

// Derived twiddle tables (built on the device from a handle's cos/sin table): one set per (kind, size, device), kept for the
// process.  derived_slot() hands out the set's two pointers; the caller holds g_derived_lock from the lookup until freshly
// built tables are published, so any number of host threads and any device index are fine.
static std::mutex g_derived_lock;
static float2 **derived_slot(int kind, int key, int dev)
{
    static std::map<std::array<int, 3>, std::array<float2 *, 2>> sets;
    return sets[{kind, key, dev}].data();
}

extern "C" int llzs_fft_f32(float *data, int count, int size, const float *cs, int inverse, void *stream)
{
    if ((size == 64 || size == 256 || size == 4096) && data && cs && count >= 1 && llzs_tune(LLZS_TUNE_FFT_GENERIC) < 1) {
        // the [E][E] twiddle table is derived once per device and size from the caller's table (kept for the process)
        int dev = 0;
        LLZ_HIP_CHECK(hipGetDevice(&dev));
        const int E = size == 64 ? 8 : size == 256 ? 16 : 64;
        std::unique_lock<std::mutex> guard(g_derived_lock);
        float2 **set = derived_slot(1, size, dev);
        if (!set[0]) {
            float2 *t = nullptr;
            LLZ_HIP_CHECK(hipMalloc(&t, sizeof(float2) * (size_t)size));
            hipLaunchKernelGGL(k_fft_square_table, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, as_stream(stream), t,
                               cs, E);
            LLZ_LAUNCH_CHECK("k_fft_square_table");
            LLZ_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));   // published only once complete: other streams may use it
            set[0] = t;
        }
        const float2 *tw2d = set[0];
        guard.unlock();
        const unsigned blocks = (unsigned)((count + (256 / E) - 1) / (256 / E));
        if (E == 8) {
            if (inverse) hipLaunchKernelGGL((k_fft_square_f32<8, true>), dim3(blocks), dim3(256), 0, as_stream(stream), data, count, tw2d);
            else hipLaunchKernelGGL((k_fft_square_f32<8, false>), dim3(blocks), dim3(256), 0, as_stream(stream), data, count, tw2d);
        } else if (E == 16) {
            if (inverse) hipLaunchKernelGGL((k_fft_square_f32<16, true>), dim3(blocks), dim3(256), 0, as_stream(stream), data, count, tw2d);
            else hipLaunchKernelGGL((k_fft_square_f32<16, false>), dim3(blocks), dim3(256), 0, as_stream(stream), data, count, tw2d);
        } else {
            if (inverse) hipLaunchKernelGGL((k_fft_square_f32<64, true>), dim3(blocks), dim3(256), 0, as_stream(stream), data, count, tw2d);
            else hipLaunchKernelGGL((k_fft_square_f32<64, false>), dim3(blocks), dim3(256), 0, as_stream(stream), data, count, tw2d);
        }
        LLZ_LAUNCH_CHECK("k_fft_square_f32");
        return LLZ_OK;
    }
    if ((size == 128 || size == 512 || size == 2048) && data && cs && count >= 1 && llzs_tune(LLZS_TUNE_FFT_GENERIC) < 1) {
        int dev = 0;
        LLZ_HIP_CHECK(hipGetDevice(&dev));
        const int E = size == 128 ? 8 : size == 512 ? 16 : 32, H = E * E;
        std::unique_lock<std::mutex> guard(g_derived_lock);
        float2 **set = derived_slot(2, size, dev);
        if (!set[0]) {
            float2 *a = nullptr, *b = nullptr;
            LLZ_HIP_CHECK(hipMalloc(&a, sizeof(float2) * (size_t)H));
            LLZ_HIP_CHECK(hipMalloc(&b, sizeof(float2) * (size_t)H));
            hipLaunchKernelGGL(k_fft_2xsquare_tables, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, as_stream(stream), a,
                               b, cs, E);
            LLZ_LAUNCH_CHECK("k_fft_2xsquare_tables");
            LLZ_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
            set[1] = b;
            set[0] = a;
        }
        const float2 *tw2d = set[0], *tw1 = set[1];
        guard.unlock();
        const unsigned blocks = (unsigned)((count + (256 / E) - 1) / (256 / E));
        if (E == 8) {
            if (inverse) hipLaunchKernelGGL((k_fft_2xsquare_f32<8, true>), dim3(blocks), dim3(256), 0, as_stream(stream), data, count, tw2d, tw1);
            else hipLaunchKernelGGL((k_fft_2xsquare_f32<8, false>), dim3(blocks), dim3(256), 0, as_stream(stream), data, count, tw2d, tw1);
        } else if (E == 16) {
            if (inverse) hipLaunchKernelGGL((k_fft_2xsquare_f32<16, true>), dim3(blocks), dim3(256), 0, as_stream(stream), data, count, tw2d, tw1);
            else hipLaunchKernelGGL((k_fft_2xsquare_f32<16, false>), dim3(blocks), dim3(256), 0, as_stream(stream), data, count, tw2d, tw1);
        } else {
            if (inverse) hipLaunchKernelGGL((k_fft_2xsquare_f32<32, true>), dim3(blocks), dim3(256), 0, as_stream(stream), data, count, tw2d, tw1);
            else hipLaunchKernelGGL((k_fft_2xsquare_f32<32, false>), dim3(blocks), dim3(256), 0, as_stream(stream), data, count, tw2d, tw1);
        }
        LLZ_LAUNCH_CHECK("k_fft_2xsquare_f32");
        return LLZ_OK;
    }
    if (size == 1024 && data && cs && count >= 1 && llzs_tune(LLZS_TUNE_FFT_GENERIC) < 1) {
        const unsigned blocks = (unsigned)((count + 7) / 8);
        if (inverse)
            hipLaunchKernelGGL(k_fft1024_f32<true>, dim3(blocks), dim3(256), 0, as_stream(stream),
                               data, count, cs);
        else
            hipLaunchKernelGGL(k_fft1024_f32<false>, dim3(blocks), dim3(256), 0, as_stream(stream),
                               data, count, cs);
        LLZ_LAUNCH_CHECK("k_fft1024_f32");
        return LLZ_OK;
    }
    return launch_fft<arith_f32>(data, count, size, cs, inverse, stream, "k_fft_radix2<f32>");
}

// MDCT frames on the register transforms (see k_mdct_reg_f32).  N in {256, 512, 1024, 2048, 4096, 8192}; cs: the FFT
// table of size N/4 the twiddle tables are derived from once per device and size.  Returns LLZ_ERR_RANGE for other N.
static int mdct_reg_launch(const float *in, float *out, int count, int N, const float *tc, const float *ts,
                           const float *cs, int inverse, void *stream, const mdct_fr *frp)
{
    int E = 0, two = 0;
    switch (N) {
    case 256: E = 8; break;  case 1024: E = 16; break; case 4096: E = 32; break;
    case 512: E = 8; two = 1; break; case 2048: E = 16; two = 1; break; case 8192: E = 32; two = 1; break;
    default: return LLZ_ERR_RANGE;
    }
    if (!in || !out || !tc || !ts || !cs || count < 1) {
        llzs_set_error("mdct4_reg_f32: bad arguments");
        return LLZ_ERR_ARG;
    }
    int dev = 0;
    LLZ_HIP_CHECK(hipGetDevice(&dev));
    const int H = E * E;
    std::unique_lock<std::mutex> guard(g_derived_lock);
    float2 **set = derived_slot(3, 2 * E + (two ? 1 : 0), dev);
    if (!set[0]) {
        float2 *a = nullptr, *b = nullptr;
        LLZ_HIP_CHECK(hipMalloc(&a, sizeof(float2) * (size_t)H));
        if (two) {
            LLZ_HIP_CHECK(hipMalloc(&b, sizeof(float2) * (size_t)H));
            hipLaunchKernelGGL(k_fft_2xsquare_tables, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, as_stream(stream), a,
                               b, cs, E);
        } else {
            hipLaunchKernelGGL(k_fft_square_table, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, as_stream(stream), a, cs,
                               E);
        }
        LLZ_LAUNCH_CHECK("mdct twiddle tables");
        LLZ_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));   // published only once complete
        set[1] = b;
        set[0] = a;
    }
    const float2 *tw2d = set[0], *tw1 = set[1];
    guard.unlock();
    const unsigned blocks = (unsigned)((count + (256 / E) - 1) / (256 / E));
    const float sqrt_cof = (float)(1.0 / sqrt((double)N));
    const mdct_fr fr = frp ? *frp : mdct_fr{};
#define LLZ_MDCT_LAUNCH(EE, TT, II)                                                                                  \
    do {                                                                                                             \
        if (frp && frp->run > 0)                                                                                     \
            hipLaunchKernelGGL((k_mdct_reg_f32<EE, TT, II, (II && (EE == 8 || (EE == 16 && !TT))) ? 2 : 1>), dim3(blocks), dim3(256), 0,        \
                               as_stream(stream), in, out, count, tc, ts, tw2d, tw1, sqrt_cof, fr);                  \
        else if (frp) hipLaunchKernelGGL((k_mdct_reg_f32<EE, TT, II, 1>), dim3(blocks), dim3(256), 0, as_stream(stream), in, \
                                         out, count, tc, ts, tw2d, tw1, sqrt_cof, fr);                               \
        else hipLaunchKernelGGL((k_mdct_reg_f32<EE, TT, II, 0>), dim3(blocks), dim3(256), 0, as_stream(stream), in,    \
                                out, count, tc, ts, tw2d, tw1, sqrt_cof, fr);                                        \
    } while (0)
#define LLZ_MDCT_PICK(EE)                                                                                            \
    do {                                                                                                             \
        if (two) { if (inverse) LLZ_MDCT_LAUNCH(EE, true, true); else LLZ_MDCT_LAUNCH(EE, true, false); }            \
        else { if (inverse) LLZ_MDCT_LAUNCH(EE, false, true); else LLZ_MDCT_LAUNCH(EE, false, false); }              \
    } while (0)
    if (E == 8) LLZ_MDCT_PICK(8);
    else if (E == 16) LLZ_MDCT_PICK(16);
    else LLZ_MDCT_PICK(32);
#undef LLZ_MDCT_PICK
#undef LLZ_MDCT_LAUNCH
    LLZ_LAUNCH_CHECK("k_mdct_reg_f32");
    return LLZ_OK;
}

extern "C" int llzs_mdct4_reg_f32(const float *in, float *out, int count, int N, const float *tc, const float *ts,
                                  const float *cs, int inverse, void *stream)
{
    return mdct_reg_launch(in, out, count, N, tc, ts, cs, inverse, stream, nullptr);
}

// Windowed 50 %-overlap MDCT frames in batch (k_mdct_reg_f32<..., FRAMES>): analysis x [channels][frames F] -> X
// [channels][frames][F], synthesis the other way with overlap-add; F = N/2, win [N], state_in / state_out [channels][F]
// (analysis: the previous frame; synthesis: the overlap-add tail), two different buffers.
extern "C" int llzs_mdct4_frames_f32(const float *in, float *out, int channels, int frames, int N, const float *tc,
                                     const float *ts, const float *cs, const float *win, const float *state_in,
                                     float *state_out, int inverse, void *stream)
{
    if (!win || !state_in || !state_out || state_in == state_out || channels < 1 || frames < 1) {
        llzs_set_error("mdct4_frames_f32: bad arguments");
        return LLZ_ERR_ARG;
    }
    mdct_fr fr;
    fr.win = win; fr.state_in = state_in; fr.state_out = state_out; fr.frames = frames; fr.run = 0;
    if (!inverse) {
        fr.first = 0; fr.step = 1; fr.per_channel = frames;
        return mdct_reg_launch(in, out, channels * frames, N, tc, ts, cs, 0, stream, &fr);
    }
    // A group per run of consecutive segments (frame lengths up to 512: the tail lives in registers; at 1024 the kernel needs
    // all 256 VGPRs and loses to the two launches, 1.03 against 0.92 ms): every output is written
    // once, finished; the frame in front of a run is transformed a second time for its tail, so runs are as long as the
    // machine stays filled with (about 64 K groups), at most 16.  Short problems keep the two launches below.
    int run = llzs_tune(LLZS_TUNE_MDCT_RUN);
    if (run < 0) {
        const long all = (long)channels * frames;
        run = all >= 4 * 65536 ? (int)(all / 65536 > 16 ? 16 : all / 65536) : 0;
    }
    if (run > 0 && N <= 1024) {
        fr.first = 0; fr.step = 1; fr.run = run; fr.per_channel = (frames + run - 1) / run;
        return mdct_reg_launch(in, out, channels * fr.per_channel, N, tc, ts, cs, 1, stream, &fr);
    }
    fr.first = 0; fr.step = 2; fr.per_channel = (frames + 1) / 2;
    int rc = mdct_reg_launch(in, out, channels * fr.per_channel, N, tc, ts, cs, 1, stream, &fr);
    if (rc == LLZ_OK && frames > 1) {
        fr.first = 1; fr.per_channel = frames / 2;
        rc = mdct_reg_launch(in, out, channels * fr.per_channel, N, tc, ts, cs, 1, stream, &fr);
    }
    return rc;
}

extern "C" int llzs_fft_f64(double *data, int size, const double *cs, int inverse, void *stream)
{
    return launch_fft<arith_f64>(data, 1, size, cs, inverse, stream, "k_fft_radix2<f64>");
}

extern "C" int llzs_fft_fixed(int *data, int count, int size, const short *cs, int inverse, void *stream)
{
    if (data && cs && count >= 1 && llzs_tune(LLZS_TUNE_FFT_GENERIC) < 1) {
#define LLZ_Q15_REG(EE, TT)                                                                                          \
    do {                                                                                                             \
        const unsigned blocks = (unsigned)((count + (256 / EE) - 1) / (256 / EE));                                   \
        if (inverse) hipLaunchKernelGGL((k_fft_reg_q15<EE, TT, true>), dim3(blocks), dim3(256), 0, as_stream(stream),  \
                                        data, count, cs);                                                            \
        else hipLaunchKernelGGL((k_fft_reg_q15<EE, TT, false>), dim3(blocks), dim3(256), 0, as_stream(stream), data,  \
                                count, cs);                                                                          \
        LLZ_LAUNCH_CHECK("k_fft_reg_q15");                                                                           \
        return LLZ_OK;                                                                                               \
    } while (0)
        switch (size) {
        case 64: LLZ_Q15_REG(8, false);
        case 128: LLZ_Q15_REG(8, true);
        case 256: LLZ_Q15_REG(16, false);
        case 512: LLZ_Q15_REG(16, true);
        case 1024: LLZ_Q15_REG(32, false);
        case 2048: LLZ_Q15_REG(32, true);
        case 4096: LLZ_Q15_REG(64, false);
        default: break;
        }
#undef LLZ_Q15_REG
    }
    return launch_fft<arith_q15>(data, count, size, cs, inverse, stream, "k_fft_radix2<q15>");
}

// fused FFT autocorrelation of `frames` frames of n float32 samples: fft length size = 2^ceil(log2(2n)) <= 4096
extern "C" int llzs_acf_fused_f32(const float *x, float *r, int frames, int n, int p, int size, const float *cs,
                                  void *stream)
{
    int log2n = 0;
    while ((1 << log2n) < size) log2n++;
    if (!x || !r || !cs || frames < 1 || n < 1 || p < 0 || p >= size || size < 8 || size > 4096 ||
        (1 << log2n) != size || 2 * n > size) {
        llzs_set_error("acf_fused_f32: bad arguments (n=%d p=%d size=%d)", n, p, size);
        return LLZ_ERR_ARG;
    }
    if ((size == 128 || size == 512) && llzs_tune(LLZS_TUNE_FFT_GENERIC) < 1) {   // the same on square_core (E = 8, 16)
        int dev = 0;
        LLZ_HIP_CHECK(hipGetDevice(&dev));
        const int E = size == 128 ? 8 : 16, H = E * E;
        std::unique_lock<std::mutex> guard(g_derived_lock);
        float2 **set = derived_slot(4, size, dev);
        if (!set[0]) {
            float2 *a = nullptr;
            LLZ_HIP_CHECK(hipMalloc(&a, sizeof(float2) * (size_t)H));
            hipLaunchKernelGGL(k_acf_sq_table, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, as_stream(stream), a, cs, E);
            LLZ_LAUNCH_CHECK("k_acf_sq_table");
            LLZ_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
            set[0] = a;
        }
        const float2 *sq = set[0];
        guard.unlock();
        const unsigned blocks = (unsigned)((frames + (256 / E) - 1) / (256 / E));
        if (E == 8) hipLaunchKernelGGL(k_acf_sq_f32<8>, dim3(blocks), dim3(256), 0, as_stream(stream), x, r, frames, n, p, sq, cs);
        else hipLaunchKernelGGL(k_acf_sq_f32<16>, dim3(blocks), dim3(256), 0, as_stream(stream), x, r, frames, n, p, sq, cs);
        LLZ_LAUNCH_CHECK("k_acf_sq_f32");
        return LLZ_OK;
    }
    if (size == 1024 && llzs_tune(LLZS_TUNE_FFT_GENERIC) < 1) {
        hipLaunchKernelGGL(k_acf1024_f32, dim3((unsigned)((frames + 7) / 8)), dim3(256), 0, as_stream(stream), x, r, frames,
                           n, p, cs);
        LLZ_LAUNCH_CHECK("k_acf1024_f32");
        return LLZ_OK;
    }
    if (size == 2048 && llzs_tune(LLZS_TUNE_FFT_GENERIC) < 1) {              // two real 2048-point transforms = two complex 1024-point ones
        hipLaunchKernelGGL(k_acf2048_f32, dim3((unsigned)((frames + 7) / 8)), dim3(256), 0, as_stream(stream), x, r,
                           frames, n, p, cs);
        LLZ_LAUNCH_CHECK("k_acf2048_f32");
        return LLZ_OK;
    }
    if (size == 4096 && llzs_tune(LLZS_TUNE_FFT_GENERIC) < 1) {              // one complex 2048-point transform on a whole wave
        hipLaunchKernelGGL(k_acf4096_f32, dim3((unsigned)((frames + 3) / 4)), dim3(256), 0, as_stream(stream), x, r,
                           frames, n, p, cs);
        LLZ_LAUNCH_CHECK("k_acf4096_f32");
        return LLZ_OK;
    }
    const int passes = (log2n + 3) / 4;
    unsigned groups = 0;
    for (int q = 0, left = log2n; q < passes; q++) {
        const int G = (left + (passes - q) - 1) / (passes - q);
        groups |= (unsigned)G << (4 * q);
        left -= G;
    }
    int tpw = 2048 / size;
    if (tpw < 1) tpw = 1;
    if (tpw > frames) tpw = frames;
    const int tstride = size + (size >> 5) + 1;
    const size_t lds = (size_t)tpw * tstride * 2 * sizeof(float) + (size_t)tw_entries(size) * 2 * sizeof(float);
    const unsigned blocks = (unsigned)((frames + tpw - 1) / tpw);
    if (lds >= 64 * 1024)
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_acf_fused_f32),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_acf_fused_f32, dim3(blocks), dim3(FFT_THREADS), lds, as_stream(stream), x, r, frames, n, p,
                       size, log2n, cs, tpw, groups);
    LLZ_LAUNCH_CHECK("k_acf_fused_f32");
    return LLZ_OK;
}

static unsigned stft_groups(int log2n)
{
    const int passes = (log2n + 3) / 4;
    unsigned groups = 0;
    for (int q = 0, left = log2n; q < passes; q++) {
        const int G = (left + (passes - q) - 1) / (passes - q);
        groups |= (unsigned)G << (4 * q);
        left -= G;
    }
    return groups;
}

static int stft_check(int channels, int frames, int F, int size, int *log2n, const char *who)
{
    *log2n = 0;
    while ((1 << *log2n) < size) (*log2n)++;
    if (channels < 1 || frames < 1 || F < 1 || size < 8 || size > 2048 || (1 << *log2n) != size ||
        (size != 2 * F && size != 4 * F)) {
        llzs_set_error("%s: bad shape (channels=%d frames=%d frame_len=%d fft_len=%d; fft_len a power of two in 8..2048)",
                       who, channels, frames, F, size);
        return LLZ_ERR_ARG;
    }
    return LLZ_OK;
}

// twiddle tables of the lane-group kernels for fft_len 256 (E = 16), 512 (2 x 16^2), 2048 (2 x 32^2), derived once per
// device from the handle's table cs
static int stft_reg_tables(int size, const float *cs, void *stream, const float2 **tw2d, const float2 **tw1)
{
    int dev = 0;
    LLZ_HIP_CHECK(hipGetDevice(&dev));
    const int E = size == 2048 ? 32 : 16, H = E * E;
    std::lock_guard<std::mutex> guard(g_derived_lock);
    float2 **set = derived_slot(5, size, dev);
    if (!set[0]) {
        float2 *a = nullptr, *b = nullptr;
        LLZ_HIP_CHECK(hipMalloc(&a, sizeof(float2) * (size_t)H));
        if (size == 256) {
            hipLaunchKernelGGL(k_fft_square_table, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, as_stream(stream), a, cs,
                               E);
        } else {
            LLZ_HIP_CHECK(hipMalloc(&b, sizeof(float2) * (size_t)H));
            hipLaunchKernelGGL(k_fft_2xsquare_tables, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, as_stream(stream), a,
                               b, cs, E);
        }
        LLZ_LAUNCH_CHECK("stft twiddle tables");
        LLZ_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
        set[1] = b;
        set[0] = a;
    }
    *tw2d = set[0];
    *tw1 = set[1];
    return LLZ_OK;
}

extern "C" int llzs_stft_analysis_f32(const float *x, const float *hist, float *re, float *im, const float *w,
                                      const float *cs, int channels, int frames, int F, int size, long x_pitch,
                                      void *stream)
{
    int log2n;
    if (!x || !hist || !re || !im || !w || !cs || x_pitch < (long)frames * F) {
        llzs_set_error("stft_analysis_f32: bad arguments");
        return LLZ_ERR_ARG;
    }
    const int rc = stft_check(channels, frames, F, size, &log2n, "stft_analysis_f32");
    if (rc != LLZ_OK) return rc;
    const long total_tr = (long)channels * frames;
    if ((size == 256 || size == 512 || size == 2048) && llzs_tune(LLZS_TUNE_FFT_GENERIC) < 1) {
        const float2 *tw2d = nullptr, *tw1 = nullptr;
        const int trc = stft_reg_tables(size, cs, stream, &tw2d, &tw1);
        if (trc != LLZ_OK) return trc;
        const int E = size == 2048 ? 32 : 16;
        const unsigned blocks = (unsigned)((total_tr + (256 / E) - 1) / (256 / E));
        if (size == 256)
            hipLaunchKernelGGL((k_stft_analysis_reg_f32<16, false>), dim3(blocks), dim3(256), 0, as_stream(stream), x, hist,
                               re, im, w, frames, F, tw2d, tw1, x_pitch, total_tr);
        else if (size == 512)
            hipLaunchKernelGGL((k_stft_analysis_reg_f32<16, true>), dim3(blocks), dim3(256), 0, as_stream(stream), x, hist, re,
                               im, w, frames, F, tw2d, tw1, x_pitch, total_tr);
        else
            hipLaunchKernelGGL((k_stft_analysis_reg_f32<32, true>), dim3(blocks), dim3(256), 0, as_stream(stream), x, hist, re,
                               im, w, frames, F, tw2d, tw1, x_pitch, total_tr);
        LLZ_LAUNCH_CHECK("k_stft_analysis_reg_f32");
        return LLZ_OK;
    }
    if (size == 1024 && llzs_tune(LLZS_TUNE_FFT_GENERIC) < 1) {
        hipLaunchKernelGGL(k_stft_analysis1024_f32, dim3((unsigned)((total_tr + 7) / 8)), dim3(256), 0, as_stream(stream),
                           x, hist, re, im, w, frames, F, cs, x_pitch, total_tr);
        LLZ_LAUNCH_CHECK("k_stft_analysis1024_f32");
        return LLZ_OK;
    }
    int tpw = 2048 / size;
    if (tpw > total_tr) tpw = (int)total_tr;
    const int tstride = size + (size >> 5) + 1;
    const size_t lds = (size_t)tpw * tstride * 2 * sizeof(float) + (size_t)tw_entries(size) * 2 * sizeof(float);
    const long blocks = (total_tr + tpw - 1) / tpw;
    hipLaunchKernelGGL(k_stft_analysis_f32, dim3((unsigned)blocks), dim3(FFT_THREADS), lds, as_stream(stream), x, hist,
                       re, im, w, frames, F, size, log2n, cs, tpw, stft_groups(log2n), x_pitch, total_tr);
    LLZ_LAUNCH_CHECK("k_stft_analysis_f32");
    return LLZ_OK;
}

extern "C" int llzs_stft_synthesis_f32(const float *re, const float *im, float *x, const float *ola_old, float *ola_new,
                                       const float *w, const float *cs, int channels, int frames, int F, int size,
                                       long x_pitch, float magic, void *stream)
{
    int log2n;
    if (!re || !im || !x || !ola_old || !ola_new || ola_old == ola_new || !w || !cs || x_pitch < (long)frames * F) {
        llzs_set_error("stft_synthesis_f32: bad arguments");
        return LLZ_ERR_ARG;
    }
    const int rc = stft_check(channels, frames, F, size, &log2n, "stft_synthesis_f32");
    if (rc != LLZ_OK) return rc;
    const int R = size / F;
    int tpw = 2048 / size;
    if (tpw > frames) tpw = frames;
    // blocks per workgroup: enough workgroups to fill the chip, long enough that the R-1 warm-up frames stay cheap
    long want = ((long)frames * channels + 2047) / 2048;
    int run_len = (int)(want < 8 * R ? 8 * R : want);
    if (run_len < tpw) run_len = tpw;
    if (run_len > frames) run_len = frames;
    if ((size == 256 || size == 512 || size == 2048) && llzs_tune(LLZS_TUNE_FFT_GENERIC) < 1) {
        const float2 *tw2d = nullptr, *tw1 = nullptr;
        const int trc = stft_reg_tables(size, cs, stream, &tw2d, &tw1);
        if (trc != LLZ_OK) return trc;
        const int gt = size == 2048 ? 8 : 16;                           // frames per group
        if (run_len < gt * R) run_len = gt * R;
        if (run_len > frames) run_len = frames;
        const int runsr = (frames + run_len - 1) / run_len;
        const dim3 grid((unsigned)((long)channels * runsr));
        // (fft_len 512 and 2048: the half-size form; tw2d of those sizes IS the E^2-point table derived from the 2 E^2-point one)
        const bool full = llzs_tune(LLZS_TUNE_STFT_FULL) == 1;
        if (size == 256)
            hipLaunchKernelGGL((k_stft_synthesis_reg_f32<16, false, false>), grid, dim3(256), 0, as_stream(stream), re, im, x,
                               ola_old, ola_new, w, frames, F, tw2d, tw1, x_pitch, run_len, runsr, magic, cs);
        else if (size == 512 && full)
            hipLaunchKernelGGL((k_stft_synthesis_reg_f32<16, true, false>), grid, dim3(256), 0, as_stream(stream), re, im, x,
                               ola_old, ola_new, w, frames, F, tw2d, tw1, x_pitch, run_len, runsr, magic, cs);
        else if (size == 512)
            hipLaunchKernelGGL((k_stft_synthesis_reg_f32<16, false, true>), grid, dim3(256), 0, as_stream(stream), re, im, x,
                               ola_old, ola_new, w, frames, F, tw2d, tw1, x_pitch, run_len, runsr, magic, cs);
        else if (full)
            hipLaunchKernelGGL((k_stft_synthesis_reg_f32<32, true, false>), grid, dim3(256), 0, as_stream(stream), re, im, x,
                               ola_old, ola_new, w, frames, F, tw2d, tw1, x_pitch, run_len, runsr, magic, cs);
        else
            hipLaunchKernelGGL((k_stft_synthesis_reg_f32<32, false, true>), grid, dim3(256), 0, as_stream(stream), re, im, x,
                               ola_old, ola_new, w, frames, F, tw2d, tw1, x_pitch, run_len, runsr, magic, cs);
        LLZ_LAUNCH_CHECK("k_stft_synthesis_reg_f32");
        return LLZ_OK;
    }
    if (size == 1024 && llzs_tune(LLZS_TUNE_FFT_GENERIC) < 1) {
        if (run_len < 8 * R) run_len = 8 * R;
        if (run_len > frames) run_len = frames;
        const int runs1k = (frames + run_len - 1) / run_len;
        hipLaunchKernelGGL(k_stft_synthesis1024_f32, dim3((unsigned)((long)channels * runs1k)), dim3(256), 0,
                           as_stream(stream), re, im, x, ola_old, ola_new, w, frames, F, cs, x_pitch, run_len, runs1k,
                           magic);
        LLZ_LAUNCH_CHECK("k_stft_synthesis1024_f32");
        return LLZ_OK;
    }
    const int runs = (frames + run_len - 1) / run_len;
    const int tstride = size + (size >> 5) + 1;
    const size_t lds = (size_t)tpw * tstride * 2 * sizeof(float) + (size_t)tw_entries(size) * 2 * sizeof(float) +
                       (size_t)(size - F) * sizeof(float);
    hipLaunchKernelGGL(k_stft_synthesis_f32, dim3((unsigned)((long)channels * runs)), dim3(FFT_THREADS), lds,
                       as_stream(stream), re, im, x, ola_old, ola_new, w, frames, F, size, log2n, cs, tpw,
                       stft_groups(log2n), x_pitch, run_len, runs, magic);
    LLZ_LAUNCH_CHECK("k_stft_synthesis_f32");
    return LLZ_OK;
}
