// fft_core.hpp -- device code shared by the FFT-based kernels (fft.hip, mdct.hip): complex type, the three arithmetic
// flavours, the LDS image and twiddle-table layout, and the fused radix-2 passes.  See fft.hip for the dataflow.
#pragma once
#include "common.hpp"
#include <type_traits>
#include <utility>

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
        // llz_fft_fixed.h:67.  (Two 24-bit multiplies on the halves of `a` -- 2 ah b + floor(al b / 2^15), exact -- are full
        // rate where this 32 x 32 low / high pair is quarter rate; measured no faster in any Q15 kernel, so the plain form stays.)
        return (int)(((long long)a * (long long)b) >> 15);
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

// LDS image of one transform: element i lives at i + i/32 (one pad element per 32) so that the strided walks of the
// late passes and the bit-reversed gather spread over the banks; transforms of a workgroup follow each other.
__device__ __forceinline__ int fft_phys(int i) { return i + (i >> 5); }

// The twiddle table (cos, sin of 2 pi i / size, i < size/2: the values of llz_fft_init / llz_fft_fixed_init, built on
// the host) is copied into LDS once per workgroup: a butterfly's twiddle is then one LDS read instead of two dword
// gathers through the vector memory path, which is what the first version of these kernels was bound by (a 1024-point
// transform made ~10,000 such gathers).  Lanes of a pass read entries a power-of-two stride apart: one pad entry per 32
// keeps strides up to 32 conflict-free.
__device__ __forceinline__ int tw_phys(int i) { return i + (i >> 5); }
__host__ __device__ constexpr int tw_entries(int size) { return (size >> 1) + (size >> 6) + 1; }

template <typename TW>
__device__ __forceinline__ void fft_load_twiddles(cpx<TW> *tw, const TW *__restrict__ cs, int size, int tid)
{
    for (int e = tid; e < (size >> 1); e += FFT_THREADS) {
        cpx<TW> t;
        t.re = cs[e];
        t.im = cs[size + e];
        tw[tw_phys(e)] = t;
    }
}

// One pass = G consecutive radix-2 stages done in registers on E = 2^G elements per work item: G barriers fewer than
// stage-by-stage, and every butterfly still is the reference's butterfly (same operands, same operation order), so
// the double and Q15 flavours stay bit-identical to llz_fft / llz_fft_fixed.
//   forward (DIF): stages with half-span hs0, hs0/2, ...;  inverse (DIT): half-span hs0, 2 hs0, ...
//   element j of an item sits at  blk * (E * step) + j * step + r,   step = distance between the item's elements
template <typename A, int G, bool INVERSE>
__device__ __forceinline__ void fft_pass(cpx<typename A::data_t> *s, int tpw, int size, int log2n, int log2step,
                                         int tstride, const cpx<typename A::tw_t> *tw, int tid)
{
    typedef typename A::data_t T;
    constexpr int E = 1 << G;
    const int step = 1 << log2step;
    const int log2items = log2n - G;                       // items per transform
    const int items = tpw << log2items;
    for (int it = tid; it < items; it += FFT_THREADS) {
        const int tr = it >> log2items, rem = it & ((1 << log2items) - 1);
        const int r = rem & (step - 1), blk = rem >> log2step;
        cpx<T> *base = s + tr * tstride;
        const int i0 = (blk << (G + log2step)) + r;
        cpx<T> v[E];
#pragma unroll
        for (int j = 0; j < E; j++) v[j] = base[fft_phys(i0 + (j << log2step))];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int hj = INVERSE ? (1 << g) : (E >> (g + 1));          // partner distance in elements of the item
            const int log2hj = INVERSE ? g : (G - 1 - g);
            const int tshift = (log2n - 1) - (log2step + log2hj);        // twiddle step = (size/2) / (hj*step)
#pragma unroll
            for (int j = 0; j < E; j++) {
                if (j & hj) continue;
                const int q = ((j & (hj - 1)) << log2step) + r;
                const int idx = q << tshift;
                const cpx<typename A::tw_t> t = tw[tw_phys(idx)];       // (cos, sin) of 2 pi idx / size
                const typename A::tw_t wr = t.re;
                const cpx<T> u = v[j], w = v[j + hj];
                if (!INVERSE) {
                    const typename A::tw_t wi = A::neg(t.im);
                    cpx<T> x, y;
                    x.re = A::add(u.re, w.re); x.im = A::add(u.im, w.im);
                    A::rot(A::sub(u.re, w.re), A::sub(u.im, w.im), wr, wi, y.re, y.im);
                    v[j] = x; v[j + hj] = y;
                } else {
                    const typename A::tw_t wi = t.im;
                    T dr, di;
                    A::rot(w.re, w.im, wr, wi, dr, di);
                    cpx<T> x, y;
                    x.re = A::add(u.re, dr); x.im = A::add(u.im, di);
                    y.re = A::sub(u.re, dr); y.im = A::sub(u.im, di);
                    v[j] = x; v[j + hj] = y;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < E; j++) base[fft_phys(i0 + (j << log2step))] = v[j];
    }
    __syncthreads();
}

// float32 flavour of a pass (tolerance, not bit-exact).  G radix-2 stages with their stage twiddles factor into an
// E-point transform whose internal twiddles are the CONSTANTS W_16^t plus ONE multiplication per element by
// W_(E*step)^(r*k) (k = the element's frequency index inside the item, r = the item's offset inside its block):
//   forward DIF:  v <- DFT_E(v) (natural in, bit-reversed out), then v[p] *= W^(r * brev(p))
//   inverse DIT:  v[p] *= conj(W)^(r * brev(p)), then inverse DFT_E (bit-reversed in, natural out)
// which is the same linear map as fft_pass (so passes of both kinds may be mixed) at a third of the instructions:
// E - 1 table reads per item instead of G*E/2, no multiplications by 1 and -i, no index arithmetic per butterfly.
template <int T16, bool INVERSE>
__device__ __forceinline__ cpx<float> mul_w16(cpx<float> d)
{
    // d * W_16^T16, W_16 = exp(-2 pi i / 16) (conjugated for the inverse)
    constexpr float R = 0.70710678118654752440f, C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
    cpx<float> y;
    if (T16 == 0) return d;
    if (T16 == 4) {                                   // -i (forward), +i (inverse)
        y.re = INVERSE ? -d.im : d.im;
        y.im = INVERSE ? d.re : -d.re;
        return y;
    }
    if (T16 == 2) {                                   // (1 - i)/sqrt2 forward
        y.re = INVERSE ? (d.re - d.im) * R : (d.re + d.im) * R;
        y.im = INVERSE ? (d.re + d.im) * R : (d.im - d.re) * R;
        return y;
    }
    if (T16 == 6) {                                   // (-1 - i)/sqrt2 forward
        y.re = INVERSE ? (-d.re - d.im) * R : (d.im - d.re) * R;
        y.im = INVERSE ? (d.re - d.im) * R : (-d.re - d.im) * R;
        return y;
    }
    const float c = (T16 == 1) ? C1 : (T16 == 3) ? S1 : (T16 == 5) ? -S1 : -C1;     // cos(2 pi T16 / 16)
    const float sn = (T16 == 1 || T16 == 7) ? S1 : C1;                               // sin(2 pi T16 / 16)
    const float wi = INVERSE ? sn : -sn;
    y.re = __builtin_fmaf(d.re, c, -(d.im * wi));
    y.im = __builtin_fmaf(d.re, wi, d.im * c);
    return y;
}

template <int E, int g, int j, bool INVERSE>
__device__ __forceinline__ void small_bfly(cpx<float> (&v)[E])
{
    constexpr int hj = INVERSE ? (1 << g) : (E >> (g + 1));
    if constexpr ((j & hj) == 0) {
        constexpr int t16 = (j & (hj - 1)) * (8 / hj);                  // W_(2hj)^(j mod hj) in sixteenths of a turn
        const cpx<float> u = v[j], w = v[j + hj];
        if (!INVERSE) {
            cpx<float> d;
            d.re = u.re - w.re; d.im = u.im - w.im;
            v[j].re = u.re + w.re; v[j].im = u.im + w.im;
            v[j + hj] = mul_w16<t16, false>(d);
        } else {
            const cpx<float> d = mul_w16<t16, true>(w);
            v[j].re = u.re + d.re; v[j].im = u.im + d.im;
            v[j + hj].re = u.re - d.re; v[j + hj].im = u.im - d.im;
        }
    }
}

template <int E, int g, bool INVERSE, int... J>
__device__ __forceinline__ void small_stage(cpx<float> (&v)[E], std::integer_sequence<int, J...>)
{
    (small_bfly<E, g, J, INVERSE>(v), ...);
}

template <int E, bool INVERSE, int... Gs>
__device__ __forceinline__ void small_fft(cpx<float> (&v)[E], std::integer_sequence<int, Gs...>)
{
    (small_stage<E, Gs, INVERSE>(v, std::make_integer_sequence<int, E>{}), ...);
}

template <int G, bool INVERSE>
__device__ __forceinline__ void fft_pass_f32(cpx<float> *s, int tpw, int size, int log2n, int log2step, int tstride,
                                             const cpx<float> *tw, int tid)
{
    constexpr int E = 1 << G;
    const int step = 1 << log2step;
    const int log2items = log2n - G;
    const int items = tpw << log2items;
    const int tshift = log2n - G - log2step;               // W_(E*step)^m = W_size^(m << tshift)
    const int half = size >> 1;
    for (int it = tid; it < items; it += FFT_THREADS) {
        const int tr = it >> log2items, rem = it & ((1 << log2items) - 1);
        const int r = rem & (step - 1), blk = rem >> log2step;
        cpx<float> *base = s + tr * tstride;
        const int i0 = (blk << (G + log2step)) + r;
        cpx<float> v[E];
#pragma unroll
        for (int j = 0; j < E; j++) v[j] = base[fft_phys(i0 + (j << log2step))];
        const int m = r << tshift;                         // W_(E*step)^(r k) = W_size^(k m), k m < size
        auto twiddle = [&](int p) {                        // v[p] *= W^(r * brev_G(p)), conjugated for the inverse
            const int k = (int)(__brev((unsigned)p) >> (32 - G));
            int idx = k * m;
            const bool wrap = idx >= half;                 // W^(idx) = -W^(idx - size/2): the table holds half a turn
            idx = wrap ? idx - half : idx;                 // (a whole-turn table cost more in LDS than it saved: measured)
            const cpx<float> t = tw[tw_phys(idx)];
            const float c = wrap ? -t.re : t.re;
            const float sn = wrap ? -t.im : t.im;
            const float wi = INVERSE ? sn : -sn;
            const cpx<float> d = v[p];
            v[p].re = __builtin_fmaf(d.re, c, -(d.im * wi));
            v[p].im = __builtin_fmaf(d.re, wi, d.im * c);
        };
        if (INVERSE && step > 1) {
#pragma unroll
            for (int p = 1; p < E; p++) twiddle(p);
        }
        small_fft<E, INVERSE>(v, std::make_integer_sequence<int, G>{});
        if (!INVERSE && step > 1) {
#pragma unroll
            for (int p = 1; p < E; p++) twiddle(p);
        }
#pragma unroll
        for (int j = 0; j < E; j++) base[fft_phys(i0 + (j << log2step))] = v[j];
    }
    __syncthreads();
}

// the float32 flavour takes the factored pass, the exact flavours the reference's butterflies
template <typename A, int G, bool INVERSE>
__device__ __forceinline__ void fft_pass_any(cpx<typename A::data_t> *s, int tpw, int size, int log2n, int log2step,
                                             int tstride, const cpx<typename A::tw_t> *tw, int tid)
{
    if constexpr (std::is_same<A, arith_f32>::value) fft_pass_f32<G, INVERSE>(s, tpw, size, log2n, log2step, tstride, tw, tid);
    else fft_pass<A, G, INVERSE>(s, tpw, size, log2n, log2step, tstride, tw, tid);
}


// split log2n radix-2 stages into ceil(log2n/4) passes of nearly equal depth (10 -> 4+3+3, 12 -> 4+4+4, 6 -> 3+3):
// G of pass p in bits [4p, 4p+4)
static inline unsigned fft_groups(int log2n)
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

// all passes of one direction over the workgroup's ntr float32 transforms (barrier after each pass)
template <bool INVERSE>
__device__ __forceinline__ void fft_run_f32(cpx<float> *s, int ntr, int size, int log2n, int tstride,
                                            const cpx<float> *tw, unsigned groups, int tid)
{
    int done = 0;
#pragma unroll 1
    for (int pss = 0; pss < 4; pss++) {
        const int G = (groups >> (4 * pss)) & 15;
        if (G == 0) break;
        const int log2step = INVERSE ? done : (log2n - done - G);
        switch (G) {
        case 1: fft_pass_f32<1, INVERSE>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        case 2: fft_pass_f32<2, INVERSE>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        case 3: fft_pass_f32<3, INVERSE>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        default: fft_pass_f32<4, INVERSE>(s, ntr, size, log2n, log2step, tstride, tw, tid); break;
        }
        done += G;
    }
}

} // namespace
