// fft32.hpp -- the in-register 32-point FFT and the half-wave 32 x 32 transpose that make a 1024-point float32
// transform out of two register passes and one LDS round trip.  Shared by the overlap-save FIR (fir_ols.hip, where it
// was developed and is described) and the 1024-point batch FFT (fft.hip).
#pragma once
#include "common.hpp"

namespace {

struct cf {
    float x, y;
};

__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }

// a * w (CONJ = false) or a * conj(w) (CONJ = true)
template <bool CONJ>
__device__ __forceinline__ cf cmul(cf a, cf w)
{
    if (CONJ) return {__builtin_fmaf(a.y, w.y, a.x * w.x), __builtin_fmaf(-a.x, w.y, a.y * w.x)};
    return {__builtin_fmaf(-a.y, w.y, a.x * w.x), __builtin_fmaf(a.x, w.y, a.y * w.x)};
}

// cos(2*pi*q/32), q = 0..8
__device__ constexpr float kCos32[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f,
                                        0.83146961230254523708f, 0.70710678118654752440f,
                                        0.55557023301960222474f, 0.38268343236508977173f,
                                        0.19509032201612826785f, 0.0f};

// d * W32^q with W32 = exp(-2*pi*j/32) (INV: exp(+2*pi*j/32)), q in 0..15 known at compile time after unrolling
template <bool INV>
__device__ __forceinline__ cf tw32(cf d, int q)
{
    if (q == 0) return d;
    if (q == 8) return INV ? cf{-d.y, d.x} : cf{d.y, -d.x};
    constexpr float r = 0.70710678118654752440f;
    if (q == 4) return INV ? cf{(d.x - d.y) * r, (d.x + d.y) * r} : cf{(d.x + d.y) * r, (d.y - d.x) * r};
    if (q == 12) return INV ? cf{-(d.x + d.y) * r, (d.x - d.y) * r} : cf{(d.y - d.x) * r, -(d.x + d.y) * r};
    const float c = q <= 8 ? kCos32[q] : -kCos32[16 - q];
    const float s = q <= 8 ? kCos32[8 - q] : kCos32[q - 8];
    // forward: d*(c - js); inverse: d*(c + js)
    return cmul<!INV>(d, cf{c, s});
}

__device__ constexpr int brev5(int r)
{
    return ((r & 1) << 4) | ((r & 2) << 2) | (r & 4) | ((r & 8) >> 2) | ((r & 16) >> 4);
}

#ifndef LLZ_OLS_DIT
#define LLZ_OLS_DIT 1
#endif

#if !LLZ_OLS_DIT
// 32-point radix-2 decimation-in-frequency FFT on registers; natural order in, v[r] = X[brev5(r)] out (456 flop)
template <bool INV>
__device__ __forceinline__ void fft32(cf (&v)[32])
{
#pragma unroll
    for (int span = 32; span >= 2; span >>= 1) {
        const int half = span >> 1;
        const int tstep = 32 / span;
#pragma unroll
        for (int blk = 0; blk < 32; blk += span) {
#pragma unroll
            for (int q = 0; q < half; q++) {
                const cf a = v[blk + q], b = v[blk + q + half];
                v[blk + q] = cadd(a, b);
                v[blk + q + half] = tw32<INV>(csub(a, b), q * tstep);
            }
        }
    }
}
#else
// One decimation-in-time butterfly (a, b) -> (a + w b, a - w b), w = W32^q (forward) or its conjugate (INV), in the
// Linzer-Feig form: the twiddle's larger component is factored out so that a general butterfly is 6 FMAs instead of
// 4 multiplies + 6 adds:  w b = c [(b.x - t b.y) + j (b.y + t b.x)],  t = +-s/c  (or the cotangent form when |s| > |c|).
template <bool INV>
__device__ __forceinline__ void bfly_dit(cf &a, cf &b, int q)
{
    const cf A = a, B = b;
    if (q == 0) {
        a = cadd(A, B); b = csub(A, B);
        return;
    }
    if (q == 8) {                               // w = -j (forward), +j (inverse)
        const cf wb = INV ? cf{-B.y, B.x} : cf{B.y, -B.x};
        a = cadd(A, wb); b = csub(A, wb);
        return;
    }
    const float c = q <= 8 ? kCos32[q] : -kCos32[16 - q];       // cos(2 pi q / 32)
    const float s0 = q <= 8 ? kCos32[8 - q] : kCos32[q - 8];    // sin(2 pi q / 32) > 0
    const float s = INV ? s0 : -s0;                             // w = c + j s
    float p, g, f;
    if (c >= s0 || -c >= s0) {                                  // |c| >= |s|: tangent form
        const float t = s / c;
        p = __builtin_fmaf(-t, B.y, B.x);
        g = __builtin_fmaf(t, B.x, B.y);
        f = c;
    } else {                                                    // cotangent form: w b = s [(r b.x - b.y) + j (r b.y + b.x)]
        // p holds -(r b.x - b.y): a negated ADDEND needs the three-operand encoding with the constant in an SGPR, which
        // issues at about 0.6 of the rate of the constant-in-the-instruction forms (tools/ubench/valu_rate.hip)
        const float r = c / s;
        p = __builtin_fmaf(-r, B.x, B.y);
        g = __builtin_fmaf(r, B.y, B.x);
        a = cf{__builtin_fmaf(-s, p, A.x), __builtin_fmaf(s, g, A.y)};
        b = cf{__builtin_fmaf(s, p, A.x), __builtin_fmaf(-s, g, A.y)};
        return;
    }
    a = cf{__builtin_fmaf(f, p, A.x), __builtin_fmaf(f, g, A.y)};
    b = cf{__builtin_fmaf(-f, p, A.x), __builtin_fmaf(-f, g, A.y)};
}

// 32-point radix-2 decimation-in-time FFT on registers (388 flop). Same contract as the DIF form above: natural
// order in, v[r] = X[brev5(r)] out -- both permutations are register renaming.
template <bool INV>
__device__ __forceinline__ void fft32(cf (&v)[32])
{
    cf w[32];
#pragma unroll
    for (int i = 0; i < 32; i++) w[i] = v[brev5(i)];
#pragma unroll
    for (int half = 1; half <= 16; half <<= 1) {
        const int tstep = 16 / half;
#pragma unroll
        for (int blk = 0; blk < 32; blk += 2 * half) {
#pragma unroll
            for (int q = 0; q < half; q++) bfly_dit<INV>(w[blk + q], w[blk + q + half], q * tstep);
        }
    }
#pragma unroll
    for (int r = 0; r < 32; r++) v[r] = w[brev5(r)];
}
#endif

constexpr int OLS_PITCH = 33;                      // +1 float per row: column walks hit 32 distinct banks, and every
                                                   // address is lane base + immediate (an XOR swizzle would save 132 B per
                                                   // plane but costs a VALU op and a register per access: measured slower)
constexpr int OLS_XBUF = 32 * OLS_PITCH;           // floats per job transpose buffer (one plane)
__device__ __forceinline__ int xaddr(int row, int col) { return row * OLS_PITCH + col; }

// 32x32 transpose of one float plane inside a half-wave: lane l5 writes its 32 registers down a column,
// then reads its row. reg index r of the source is stored at row brev5(r) (undoing the FFT's output order).
#define OLS_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// wcol: the column this lane's registers go to (its index in the twiddle W^(row * wcol) as well); rrow: the row it reads
// back.  The half-wave's lanes must cover 0..31 once in each (any lane order: the overlap-save kernel's wide I/O form
// owns permuted columns).
template <bool INV>
__device__ __forceinline__ void transpose_twiddle(cf (&v)[32], float *buf, const float2 *__restrict__ tw, int wcol, int rrow)
{
    // inter-pass twiddle W_1024^(+-brev5(r)*wcol) applied on the way out, then two single-plane transposes
#pragma unroll
    for (int r = 0; r < 32; r++) {
        const float2 w = tw[brev5(r) * 32 + wcol];
        v[r] = cmul<INV>(v[r], cf{w.x, w.y});
    }
#pragma unroll
    for (int r = 0; r < 32; r++) buf[xaddr(brev5(r), wcol)] = v[r].x;
    OLS_WAVE_SYNC();
#pragma unroll
    for (int cidx = 0; cidx < 32; cidx++) v[cidx].x = buf[xaddr(rrow, cidx)];
    OLS_WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < 32; r++) buf[xaddr(brev5(r), wcol)] = v[r].y;
    OLS_WAVE_SYNC();
#pragma unroll
    for (int cidx = 0; cidx < 32; cidx++) v[cidx].y = buf[xaddr(rrow, cidx)];
    OLS_WAVE_SYNC();
}

template <bool INV>
__device__ __forceinline__ void transpose_twiddle(cf (&v)[32], float *buf, const float2 *__restrict__ tw, int l5)
{
    transpose_twiddle<INV>(v, buf, tw, l5, l5);
}

// x's upper half-wave <-> y's lower half-wave (the radix-2 step that splits a transform over the two half-waves of a wave:
// fir_ols.hip k_fir_ols2k_chain_f32, fft.hip k_acf4096_f32)
__device__ __forceinline__ void swap32(float &x, float &y)
{
    const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(p[0]);
    y = __uint_as_float(p[1]);
}

} // namespace
