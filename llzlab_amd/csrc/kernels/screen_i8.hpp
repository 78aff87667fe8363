// screen_i8.hpp -- the integer decision shared by the two bit-exact int16 resampling kernels (fir_mfma_i8.hip: L = 1;
// resample_i8.hip: general L/M).  Both form, per output, five int32 accumulators a_v (v = 0..4, weight 256^v) of digit-plane
// products on v_mfma_i32_16x16x64_i8 such that
//        T' = a0 + 2^8 a1 + 2^16 a2 + 2^24 a3 + 2^32 a4 + Bq,        value of the output  v = T' 2^-(32 + rs)  up to eps
// (Bq = floor(128 sum G / 256): the samples' +128 offset; what is left out -- the weight-0 product and the low 8 bits of the
// bias -- is bounded by the host and sits inside eps).
#pragma once
#include "common.hpp"

typedef int scr_i32x4 __attribute__((ext_vector_type(4)));

// lane n of a 16-lane row takes column scr_col(n) of the tile's 16, and quarter kq of the lanes takes the 16-byte chunk
// scr_chunk(kq) of a 64-sample step: when the columns are an ODD number of 16-byte chunks apart, the sixteen lanes a
// ds_read_b128 serves together ({0-3, 12-15} of one quarter with {4-11} of its neighbour) then read sixteen different
// 16-byte bank groups (odd columns against even columns + 2 chunks) -- the natural order collides two by two
__host__ __device__ __forceinline__ int scr_col(int n) { return n < 4 ? 2 * n + 1 : (n < 12 ? 2 * (n - 4) : 2 * (n - 12) + 9); }
__host__ __device__ __forceinline__ int scr_chunk(int kq) { return ((kq & 1) << 1) | (kq >> 1); }

// Start values of an output's accumulators.  The matrix cores add into whatever the accumulator registers hold, so the bias
// Bq rides along for free, and so does an offset of 2^31 that makes the low pair a0 + 2^8 a1 a NON-NEGATIVE 32-bit number
// (|a0 + 2^8 a1| < 2^30.7): with Bq - 2^31 = s0' + 2^16 s2 + 2^32 s4 (s0', s2 in 0 .. 65535), accumulator 0 starts from
// 2^31 + s0' (as a wrapped int32), 2 from s2, 4 from s4, the other two from 0.  T' is then the plain 64-bit sum
// (a4 : low pair) + (q << 16), q = a2 + 2^8 a3: no sign extension of the low pair, no separate bias addition, one carry.
__device__ __forceinline__ void scr_start(int bq_lo, int bq_hi, int &s0, int &s2, int &s4)
{
    const long long b = (long long)(((unsigned long long)(unsigned)bq_hi << 32) | (unsigned)bq_lo) - (1ll << 31);
    s0 = (int)(0x80000000u + (unsigned)(b & 0xffff));
    s2 = (int)((b >> 16) & 0xffff);
    s4 = (int)(b >> 32);
}

// One output from its five accumulators (started as scr_start says) in exact integer arithmetic (|a_v| < 2^22.7 without the
// start values: at most 200 taps per product pair): T' as a 64-bit pair (hi, lo), I = floor(T' / 2^(32 + rs)), F = the top 32
// bits of the fraction below it.  Returns the value truncated toward zero, NOT yet clamped (= the reference's result after
// scr_clamp) when no integer lies within eps of the value -- F farther than e32 from both ends -- and sets `unsure`
// otherwise.  NEG selects rs < 0.  rs in [-8, 6]; e32 = ceil(eps 2^32) + 2.  not_integer = 0: the caller knows the output IS
// an integer (then it is I itself, and `unsure` is always set: ignore it).
template <bool NEG>
__device__ __forceinline__ int scr_decide(int a0, int a1, int a2, int a3, int a4, int rs, unsigned e32, bool &unsure,
                                          unsigned not_integer = 1u, unsigned e32_twice = 0u)
{
    // (e32_twice: 2 e32 from the caller's own scalar register when e32 itself sits in a vector register)
    if (e32_twice == 0u) e32_twice = 2u * e32;
    const unsigned p = (unsigned)a0 + ((unsigned)a1 << 8);
    const int q = a2 + (a3 << 8);
    // (a4 : p) + (q << 16) as two 32-bit additions chained by the carry
    unsigned c1, c2;
    const unsigned lo = __builtin_addc(p, (unsigned)q << 16, 0u, &c1);
    const int hi = (int)__builtin_addc((unsigned)a4, (unsigned)(q >> 16), c1, &c2);
    int I;
    unsigned F;
    if (NEG) {
        I = (int)__builtin_amdgcn_alignbit((unsigned)hi, lo, (unsigned)(32 + rs));
        F = lo << (-rs);
    } else {
        I = hi >> rs;
        F = __builtin_amdgcn_alignbit((unsigned)hi, lo, (unsigned)rs);
    }
    unsure = F + e32 <= e32_twice;                              // wrapping: the fraction is within e32 of 0 or of 1
    // toward zero: the value is not an integer here, so a negative one moves up by one -- the sign bit, taken as a bit field
    // of width not_integer (0 where the caller knows the output IS an integer: then it is I itself)
    return I + (int)__builtin_amdgcn_ubfe((unsigned)I, 31u, not_integer);
}

// An undecided output whose value lies within eps of ZERO is decided after all: the reference's y then lies in (-1, 1) and
// truncates to 0 from either side -- which is what scr_decide returned (I = 0: 0; I = -1: -1 + 1).  Digital silence makes EVERY
// output such a value (all products vanish: T' = 0 exactly), so without this a silent channel takes the second look for every
// sample (measured on k_resample_i8d: 11.0 ms against 0.64 ms for noise).  Evaluated in the rare branch only.
template <bool NEG>
__device__ __forceinline__ bool scr_near_zero(int a0, int a1, int a2, int a3, int a4, int rs, unsigned e32)
{
    const unsigned p = (unsigned)a0 + ((unsigned)a1 << 8);
    const int q = a2 + (a3 << 8);
    unsigned c1, c2;
    const unsigned lo = __builtin_addc(p, (unsigned)q << 16, 0u, &c1);
    const int hi = (int)__builtin_addc((unsigned)a4, (unsigned)(q >> 16), c1, &c2);
    int I;
    unsigned F;
    if (NEG) {
        I = (int)__builtin_amdgcn_alignbit((unsigned)hi, lo, (unsigned)(32 + rs));
        F = lo << (-rs);
    } else {
        I = hi >> rs;
        F = __builtin_amdgcn_alignbit((unsigned)hi, lo, (unsigned)rs);
    }
    const unsigned t = F + e32;                                 // within e32 above 0: t in [e32, 2 e32]; below 0: t in [0, e32)
    return (I == 0 && t >= e32 && t <= 2u * e32) || (I == -1 && t < e32);
}

// the reference's clamp (llz_resample.c:596-599) on the truncated value
__device__ __forceinline__ short scr_clamp(int t)
{
    t = t > 32767 ? 32767 : t;
    t = t < -32768 ? -32768 : t;
    return (short)t;
}
// an undecided output needs the reference's own arithmetic only when its value can end up inside the clamp range or on a
// rail's edge: |truncated value| <= 32769 (beyond that the int16 is the rail whatever the last digits are)
__device__ __forceinline__ bool scr_in_reach(int t) { return (unsigned)(t + 32769) <= 65538u; }

// nine products of one 64-sample step: tap digit planes 0..4 (ad) against the high sample plane (weights 1..5 -> accumulators
// 0..4) and planes 1..4 against the low one (weights 1..4 -> accumulators 0..3); FIRST starts the accumulators from their
// start values (scr_start: accumulators 0, 2, 4; the constant 0 for 1 and 3)
template <bool FIRST>
__device__ __forceinline__ void scr_step(scr_i32x4 (&acc)[5], const scr_i32x4 (&ad)[5], const scr_i32x4 &b_lo, const scr_i32x4 &b_hi,
                                         const scr_i32x4 (&start)[3])
{
#pragma unroll
    for (int p = 0; p < 5; p++) {
        const scr_i32x4 c0 = !FIRST ? acc[p] : ((p & 1) ? (scr_i32x4){0, 0, 0, 0} : start[p >> 1]);
        acc[p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ad[p], b_hi, c0, 0, 0, 0);
        if (p > 0) acc[p - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ad[p], b_lo, acc[p - 1], 0, 0, 0);
    }
}

// four decided values of a lane, clamped (llz_resample.c:596-599: v_cvt_pk_i16_i32 saturates) and packed in memory order
typedef short scr_i16x4 __attribute__((ext_vector_type(4)));
typedef short scr_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ scr_i16x4 scr_clamp4(const int (&r)[4])
{
    const scr_i16x2 a = __builtin_amdgcn_cvt_pk_i16(r[0], r[1]), b = __builtin_amdgcn_cvt_pk_i16(r[2], r[3]);
    return (scr_i16x4){a[0], a[1], b[0], b[1]};
}
