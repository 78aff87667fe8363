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

// One output from its five accumulators in exact 32-bit integer arithmetic (|a_v| < 2^22.7: at most 200 taps per product
// pair): T' as a 64-bit pair (hi, lo) by two add-with-carry steps, I = floor(T' / 2^(32 + rs)), F = the top 32 bits of the
// fraction below it.  Returns the value truncated toward zero, NOT yet clamped (= the reference's result after scr_clamp) when
// no integer lies within eps of the value -- F farther than e32 from both ends -- and sets `unsure` otherwise.  NEG selects rs < 0.  rs in [-8, 6]; e32 = ceil(eps 2^32) + 2.
template <bool NEG>
__device__ __forceinline__ int scr_decide(int a0, int a1, int a2, int a3, int a4, int bq_lo, int bq_hi, int rs, unsigned e32,
                                          bool &unsure, bool integer_valued = false)
{
    const int p = a0 + (a1 << 8);
    const int q = a2 + (a3 << 8);
    // T' = (p + Bq) + q 2^16 + a4 2^32  (q 2^16 = (q >> 16) 2^32 + (q << 16) mod 2^32)
    unsigned c1, c2;
    const unsigned lo_w = __builtin_addc((unsigned)p, (unsigned)bq_lo, 0u, &c1);
    const unsigned hi_w = (unsigned)(p >> 31) + (unsigned)bq_hi + c1;
    const unsigned lo = __builtin_addc(lo_w, (unsigned)q << 16, 0u, &c2);
    const int hi = (int)(hi_w + (unsigned)((q >> 16) + a4) + c2);
    int I;
    unsigned F;
    if (NEG) {
        I = (int)__builtin_amdgcn_alignbit((unsigned)hi, lo, (unsigned)(32 + rs));
        F = lo << (-rs);
    } else {
        I = hi >> rs;
        F = __builtin_amdgcn_alignbit((unsigned)hi, lo, (unsigned)rs);
    }
    unsure = F + e32 <= 2u * e32;                               // wrapping: the fraction is within e32 of 0 or of 1
    // toward zero: the value is not an integer here (unless the caller knows the output IS one: then it is I itself)
    return integer_valued ? I : I + (int)((unsigned)I >> 31);
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
// 0..4) and planes 1..4 against the low one (weights 1..4 -> accumulators 0..3); FIRST starts the accumulators from 0
template <bool FIRST>
__device__ __forceinline__ void scr_step(scr_i32x4 (&acc)[5], const scr_i32x4 (&ad)[5], const scr_i32x4 &b_lo, const scr_i32x4 &b_hi)
{
#pragma unroll
    for (int p = 0; p < 5; p++) {
        const scr_i32x4 c0 = FIRST ? (scr_i32x4){0, 0, 0, 0} : acc[p];
        acc[p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ad[p], b_hi, c0, 0, 0, 0);
        if (p > 0) acc[p - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ad[p], b_lo, acc[p - 1], 0, 0, 0);
    }
}
