// fir_mfma.hip -- K1m: time-domain FIR / decimating FIR on the matrix cores of gfx950 (split-bf16, fp32-exact products).
//
//   y[c][i] = gain * sum_{k<T} taps[k] * x[c][i*M - k]        (M = 1: llz_fir.c:411-426 as driven by :570-580;
//                                                              M > 1: llz_resample.c:583-603 with L = 1)
//
// Why matrix cores for a filter.  At 126..270 flop per 5.3..8 bytes the time-domain form is bound by the fp32 VALU, not
// by HBM (fir_td.hip: 44 % of the HBM roofline at 63 taps; the LDS polyphase decimator of resample.hip: 44 %, 52 ms on
// 8192 ch x 4 Mi, T=134, M=3).  Measured steps from there, same workload:
//   * v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate) on the Toeplitz form below: 44 ms.  It issues the VALU's
//     64 flop/clk/SIMD without operand shuffling, but the chip holds only ~1.6 GHz under it: bound by the matrix pipe
//     (35 ms with the input loads removed, 33 ms with the MFMAs removed, no better with a double-buffered LDS image).
//   * the split-bf16 form in this file: 37 ms = 61 % of the 8 TB/s roofline (33 ms with the MFMAs removed: the rest is
//     phase overlap between the two resident workgroups of a CU).
//
// Split-bf16.  The bf16 MFMA is 16x faster per flop than the fp32 one, and an fp32 number is EXACTLY the sum of three
// bf16 numbers (8 + 8 + 8 mantissa bits, each the round-to-nearest bf16 of the running remainder; bf16 has fp32's
// exponent range).  With x = x1 + x2 + x3 and h = h1 + h2 + h3 the product x*h is
//     x1h1 + (x1h2 + x2h1) + (x2h2 + x1h3 + x3h1)        up to terms below 2^-26 |x h|;
// every bf16 product is exact in fp32 and the MFMA accumulates in fp32, so the result carries fp32 rounding error like
// fir_td.hip (tests: same 1e-5 RMS bar against the double oracle, measured ~1e-7).  Six bf16 MFMA terms cost 6/16 of one
// fp32 MFMA term.  Samples are split once, when a tile is written to LDS (three bf16 planes, 6 B per sample); taps once,
// when the Toeplitz table is built.
//
// Mapping.  A wave computes a 16 x 16 tile D[m][n] = output (16*seg(n) + m): column n is one of 16 consecutive
// 16-output segments of a channel, row m the output inside the segment.  With the segment's input window
// w_n[t] = x[16*seg(n)*M - tpad + t]  (t = 0 .. tpad + 15*M),  D = A * B where
//     A[m][t] = gain * taps[m*M + tpad - t]   (zero outside 0..T-1)   -- a banded Toeplitz block, the same for every tile
//     B[t][n] = w_n[t]
// and t is walked 32 at a time by v_mfma_f32_16x16x32_bf16: lane l supplies A[l&15][8(l>>4) + j] and
// B[8(l>>4) + j][l&15], j = 0..7, as one 16-byte register quad, i.e. eight CONSECUTIVE window samples of its segment:
// one ds_read_b128 per plane and step, bank-conflict free without padding (segment bases are 32*M bytes apart).
// Of the 16 x 32*ksteps entries of A, 16*T are non-zero: 70 % at T=134, M=3.
//
// Data movement.  Persistent workgroups (4 waves) walk tiles of 4*NACC*256 consecutive outputs of one channel.  The
// inputs of a tile (halo tpad) are loaded once with 16-byte coalesced loads: the loads of the NEXT tile are issued into
// registers before the MFMA phase of the current one, split and written to LDS after it.  The A table
// [3 parts][ksteps][64 lanes][8] is built once per workgroup.  Each wave runs NACC independent accumulators.  Results
// leave as one 16-byte non-temporal store per lane and accumulator: a wave writes 1 KB contiguous.
#include "common.hpp"
#include <stdlib.h>
#include <string.h>

#ifndef LLZ_MF_DIAG
#define LLZ_MF_DIAG 0       // ablation builds: 1 = no MFMA phase, 2 = no input loads, 3 = no split + LDS write
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MF_WAVES = 4;
constexpr int MF_THREADS = MF_WAVES * 64;
constexpr int MF_NV_MAX = 16;              // prefetch registers: 16 x 16 B per thread = at most 16 Ki floats per tile

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct mfb_shape {
    int T, M;
    int tpad;        // multiple of 4 >= T-1
    int ksteps;      // MFMA steps of 32 window samples: ceil((tpad + 15*M + 1) / 32)
    int total;       // samples staged per tile (multiple of 8)
    int plane;       // elements per LDS plane (multiple of 8)
    int tiles_per_ch;
};

__device__ __forceinline__ void mfb_split(float x, __bf16 &b1, __bf16 &b2, __bf16 &b3)
{
    b1 = (__bf16)x;
    float r = x - (float)b1;                             // exact
    b2 = (__bf16)r;
    r -= (float)b2;                                      // exact
    b3 = (__bf16)r;
}

template <int NACC, int MF_NV>
__global__ void __launch_bounds__(MF_THREADS)
k_fir_mfma_bf16x3(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
                  const float *__restrict__ taps, long n_in, long n_out, long in_pitch, long out_pitch, float gain,
                  mfb_shape sh, long ntiles)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TILE_OUT = MF_WAVES * NACC * 256;
    const int aplane = sh.ksteps * 512;                  // A table: [3 parts][ksteps][64 lanes][8]
    __bf16 *atab = reinterpret_cast<__bf16 *>(lds);
    __bf16 *xs = atab + 3 * aplane;                      // input image: [3 parts][plane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int total = sh.total, last4 = sh.total - 4, plane = sh.plane;
    const bool aligned_in = (in_pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const bool aligned_out = (out_pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;

    for (int e = tid; e < aplane; e += MF_THREADS) {
        const int s = e >> 9, l = (e >> 3) & 63, j = e & 7;
        const int t = 32 * s + 8 * (l >> 4) + j;
        const int k = (l & 15) * sh.M + sh.tpad - t;
        const float h = (k >= 0 && k < sh.T) ? taps[k] * gain : 0.f;
        mfb_split(h, atab[e], atab[aplane + e], atab[2 * aplane + e]);
    }

    auto tile_first = [&](long q, int &c, long &o0) {
        c = (int)(q / sh.tiles_per_ch);
        o0 = (q - (long)c * sh.tiles_per_ch) * TILE_OUT;
        return o0 * sh.M - sh.tpad;
    };
    auto is_interior = [&](long first) { return aligned_in && first >= 0 && first + total <= n_in; };

    auto prefetch = [&](f32x4 (&v)[MF_NV], long q) {
        int c; long o0;
        const long first = tile_first(q, c, o0);
        if (LLZ_MF_DIAG == 2 || !is_interior(first)) return false;
        const float *src = in + (size_t)c * in_pitch + first;
#pragma unroll
        for (int j = 0; j < MF_NV; j++) {
            if (j * MF_THREADS * 4 < total) {
                int p = (j * MF_THREADS + tid) * 4;
                p = p < last4 ? p : last4;
                v[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + p));
            }
        }
        return true;
    };

    auto tile = [&](long q, f32x4 (&v)[MF_NV], bool &have) {
        int c; long o0;
        const long first = tile_first(q, c, o0);
        __syncthreads();
        if (have) {
#pragma unroll
            for (int j = 0; j < MF_NV; j++) {
                if (j * MF_THREADS * 4 < total) {
                    const int p = (j * MF_THREADS + tid) * 4;
                    if (LLZ_MF_DIAG == 3) { asm volatile("" :: "v"(v[j])); continue; }
                    bf16x4 p1, p2, p3;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        __bf16 b1, b2, b3;
                        mfb_split(v[j][i], b1, b2, b3);
                        p1[i] = b1; p2[i] = b2; p3[i] = b3;
                    }
                    if (p < total) {
                        *reinterpret_cast<bf16x4 *>(&xs[p]) = p1;
                        *reinterpret_cast<bf16x4 *>(&xs[plane + p]) = p2;
                        *reinterpret_cast<bf16x4 *>(&xs[2 * plane + p]) = p3;
                    }
                }
            }
        } else if (LLZ_MF_DIAG != 2) {
            const float *row = in + (size_t)c * in_pitch;
            for (int base = 0; base < total; base += 8 * MF_THREADS) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    long idx = first + base + j * MF_THREADS + tid;
                    idx = idx < 0 ? 0 : (idx < n_in ? idx : n_in - 1);
                    x[j] = row[idx];
                }
                // the uses below sit behind per-lane branches; without this the compiler's wait-count tracking carries
                // "x[] may still be in flight" into the MFMA loop and stalls it on the prefetch loads (vmcnt(0) = 0x0F70)
                __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int p = base + j * MF_THREADS + tid;
                    const long idx = first + p;
                    __bf16 b1, b2, b3;
                    mfb_split((idx >= 0 && idx < n_in) ? x[j] : 0.f, b1, b2, b3);
                    if (p < total) { xs[p] = b1; xs[plane + p] = b2; xs[2 * plane + p] = b3; }
                }
            }
            if (first < 0 && hist) {
                const float *hrow = hist + (size_t)c * (sh.T - 1);
                for (int p = tid; p < sh.tpad; p += MF_THREADS) {
                    const long idx = first + p;
                    if (idx < 0 && idx >= -(long)(sh.T - 1)) {
                        __bf16 b1, b2, b3;
                        mfb_split(hrow[sh.T - 1 + idx], b1, b2, b3);
                        xs[p] = b1; xs[plane + p] = b2; xs[2 * plane + p] = b3;
                    }
                }
            }
        }
        __syncthreads();
        have = q + gridDim.x < ntiles && prefetch(v, q + gridDim.x);

        f32x4 acc[NACC];
        const __bf16 *bp[NACC];
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
            bp[a] = xs + ((wave * NACC + a) * 16 + n) * 16 * sh.M + 8 * kq;
        }
        const __bf16 *ap = atab + lane * 8;
        if (LLZ_MF_DIAG != 1) {
            bf16x8 a0[3], a1[3], b0[NACC][3], b1[NACC][3];
            auto fetch = [&](int s, bf16x8 (&ad)[3], bf16x8 (&bd)[NACC][3]) {
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    ad[r] = *reinterpret_cast<const bf16x8 *>(ap + r * aplane + s * 512);
#pragma unroll
                    for (int a = 0; a < NACC; a++)
                        bd[a][r] = *reinterpret_cast<const bf16x8 *>(bp[a] + r * plane + s * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto mac = [&](const bf16x8 (&ad)[3], const bf16x8 (&bd)[NACC][3]) {
                // (tap part, sample part): smallest terms first
                constexpr int ta[6] = {0, 2, 1, 0, 1, 0};
                constexpr int tb[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                for (int e = 0; e < 6; e++)
#pragma unroll
                    for (int a = 0; a < NACC; a++)
                        acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad[ta[e]], bd[a][tb[e]], acc[a], 0, 0, 0);
            };
            int s = 0;
            fetch(0, a0, b0);                            // ksteps >= 1
            for (; s + 2 <= sh.ksteps; s += 2) {
                fetch(s + 1, a1, b1);
                mac(a0, b0);
                if (s + 2 < sh.ksteps) fetch(s + 2, a0, b0);
                mac(a1, b1);
            }
            if (s < sh.ksteps) mac(a0, b0);
        }

        float *orow = out + (size_t)c * out_pitch;
        const bool whole = aligned_out && o0 + TILE_OUT <= n_out;
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            const long o = o0 + ((wave * NACC + a) * 16 + n) * 16 + 4 * kq;
            if (whole) {
                __builtin_nontemporal_store(acc[a], reinterpret_cast<f32x4 *>(orow + o));
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (o + j < n_out) orow[o + j] = acc[a][j];
            }
        }
    };

    long q = blockIdx.x;
    f32x4 v[MF_NV];
    bool have = q < ntiles && prefetch(v, q);
    for (; q < ntiles; q += gridDim.x) tile(q, v, have);
}

// ---------------------------------------------------------------------------------------------------------------------
// int16 PCM in and out (the reference resampler's own sample format, llz_resample.c:583-603) on the same scheme.
// A 16-bit integer is EXACTLY the sum of two bf16 numbers (the round-to-nearest bf16 and an 8-bit remainder), so the
// input needs two planes; taps stay three parts: all six products are kept, nothing is dropped.  Accumulation is fp32
// instead of the reference's double, so this is NOT the bit-exact path (that is k_resample_i16_exact): the sum carries
// ~1e-7 relative error (about 0.01 LSB at full scale) before the reference's clamp and truncation toward zero, i.e. a
// sample differs from the reference by one LSB when its exact value sits that close to an integer -- the "within 1 LSB,
// RMS <= 1e-5 of full scale" contract of SURVEY.md 8(d), selected explicitly by the caller (LLZ_PCM_I16_FAST).
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4 __attribute__((ext_vector_type(4)));

template <int NACC, int MF_NV>
__global__ void __launch_bounds__(MF_THREADS)
k_fir_mfma_i16(const short *__restrict__ in, short *__restrict__ out, const short *__restrict__ hist,
               const float *__restrict__ taps, long n_in, long n_out, long in_pitch, long out_pitch, float gain,
               mfb_shape sh, long ntiles)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TILE_OUT = MF_WAVES * NACC * 256;
    const int aplane = sh.ksteps * 512;                  // A table: [3 parts][ksteps][64 lanes][8]
    __bf16 *atab = reinterpret_cast<__bf16 *>(lds);
    __bf16 *xs = atab + 3 * aplane;                      // input image: [2 parts][plane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int total = sh.total, last8 = sh.total - 8, plane = sh.plane;
    const bool aligned_in = (in_pitch & 7) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const bool aligned_out = (out_pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0;

    for (int e = tid; e < aplane; e += MF_THREADS) {
        const int s = e >> 9, l = (e >> 3) & 63, j = e & 7;
        const int t = 32 * s + 8 * (l >> 4) + j;
        const int k = (l & 15) * sh.M + sh.tpad - t;
        const float h = (k >= 0 && k < sh.T) ? taps[k] * gain : 0.f;
        mfb_split(h, atab[e], atab[aplane + e], atab[2 * aplane + e]);
    }

    auto split16 = [](short sv, __bf16 &b1, __bf16 &b2) {
        const float x = (float)sv;
        b1 = (__bf16)x;
        b2 = (__bf16)(x - (float)b1);                    // exact: at most 8 significant bits remain
    };
    auto tile_first = [&](long q, int &c, long &o0) {
        c = (int)(q / sh.tiles_per_ch);
        o0 = (q - (long)c * sh.tiles_per_ch) * TILE_OUT;
        return o0 * sh.M - sh.tpad;                      // tpad is a multiple of 8 here: 16-byte aligned rows of int16
    };
    auto is_interior = [&](long first) { return aligned_in && first >= 0 && first + total <= n_in; };

    auto prefetch = [&](i16x8 (&v)[MF_NV], long q) {
        int c; long o0;
        const long first = tile_first(q, c, o0);
        if (!is_interior(first)) return false;
        const short *src = in + (size_t)c * in_pitch + first;
#pragma unroll
        for (int j = 0; j < MF_NV; j++) {
            if (j * MF_THREADS * 8 < total) {
                int p = (j * MF_THREADS + tid) * 8;
                p = p < last8 ? p : last8;
                v[j] = __builtin_nontemporal_load(reinterpret_cast<const i16x8 *>(src + p));
            }
        }
        return true;
    };

    auto tile = [&](long q, i16x8 (&v)[MF_NV], bool &have) {
        int c; long o0;
        const long first = tile_first(q, c, o0);
        __syncthreads();
        if (have) {
#pragma unroll
            for (int j = 0; j < MF_NV; j++) {
                if (j * MF_THREADS * 8 < total) {
                    const int p = (j * MF_THREADS + tid) * 8;
                    bf16x8 p1, p2;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        __bf16 b1, b2;
                        split16(v[j][i], b1, b2);
                        p1[i] = b1; p2[i] = b2;
                    }
                    if (p < total) {
                        *reinterpret_cast<bf16x8 *>(&xs[p]) = p1;
                        *reinterpret_cast<bf16x8 *>(&xs[plane + p]) = p2;
                    }
                }
            }
        } else {
            const short *row = in + (size_t)c * in_pitch;
            for (int base = 0; base < total; base += 8 * MF_THREADS) {
                short x[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    long idx = first + base + j * MF_THREADS + tid;
                    idx = idx < 0 ? 0 : (idx < n_in ? idx : n_in - 1);
                    x[j] = row[idx];
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): see k_fir_mfma_bf16x3
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int p = base + j * MF_THREADS + tid;
                    const long idx = first + p;
                    __bf16 b1, b2;
                    split16((idx >= 0 && idx < n_in) ? x[j] : (short)0, b1, b2);
                    if (p < total) { xs[p] = b1; xs[plane + p] = b2; }
                }
            }
            if (first < 0 && hist) {
                const short *hrow = hist + (size_t)c * (sh.T - 1);
                for (int p = tid; p < sh.tpad; p += MF_THREADS) {
                    const long idx = first + p;
                    if (idx < 0 && idx >= -(long)(sh.T - 1)) {
                        __bf16 b1, b2;
                        split16(hrow[sh.T - 1 + idx], b1, b2);
                        xs[p] = b1; xs[plane + p] = b2;
                    }
                }
            }
        }
        __syncthreads();
        have = q + gridDim.x < ntiles && prefetch(v, q + gridDim.x);

        f32x4 acc[NACC];
        const __bf16 *bp[NACC];
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
            bp[a] = xs + ((wave * NACC + a) * 16 + n) * 16 * sh.M + 8 * kq;
        }
        const __bf16 *ap = atab + lane * 8;
        {
            bf16x8 a0[3], a1[3], b0[NACC][2], b1[NACC][2];
            auto fetch = [&](int s, bf16x8 (&ad)[3], bf16x8 (&bd)[NACC][2]) {
#pragma unroll
                for (int r = 0; r < 3; r++) ad[r] = *reinterpret_cast<const bf16x8 *>(ap + r * aplane + s * 512);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int a = 0; a < NACC; a++)
                        bd[a][r] = *reinterpret_cast<const bf16x8 *>(bp[a] + r * plane + s * 32);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto mac = [&](const bf16x8 (&ad)[3], const bf16x8 (&bd)[NACC][2]) {
                constexpr int ta[6] = {2, 1, 2, 0, 1, 0};           // (tap part, sample part): smallest terms first
                constexpr int tb[6] = {1, 1, 0, 1, 0, 0};
#pragma unroll
                for (int e = 0; e < 6; e++)
#pragma unroll
                    for (int a = 0; a < NACC; a++)
                        acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad[ta[e]], bd[a][tb[e]], acc[a], 0, 0, 0);
            };
            int s = 0;
            fetch(0, a0, b0);
            for (; s + 2 <= sh.ksteps; s += 2) {
                fetch(s + 1, a1, b1);
                mac(a0, b0);
                if (s + 2 < sh.ksteps) fetch(s + 2, a0, b0);
                mac(a1, b1);
            }
            if (s < sh.ksteps) mac(a0, b0);
        }

        short *orow = out + (size_t)c * out_pitch;
        const bool whole = aligned_out && o0 + TILE_OUT <= n_out;
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            const long o = o0 + ((wave * NACC + a) * 16 + n) * 16 + 4 * kq;
            i16x4 y;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float t = acc[a][j];                                 // the taps carry the gain (llz_resample.c:594)
                t = t > 32767.f ? 32767.f : t;                       // :596-599
                t = t < -32768.f ? -32768.f : t;
                y[j] = (short)(int)t;                                // :601, truncation toward zero
            }
            if (whole) {
                *reinterpret_cast<i16x4 *>(orow + o) = y;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (o + j < n_out) orow[o + j] = y[j];
            }
        }
    };

    long q = blockIdx.x;
    i16x8 v[MF_NV];
    bool have = q < ntiles && prefetch(v, q);
    for (; q < ntiles; q += gridDim.x) tile(q, v, have);
}

bool mfb_make_shape(int T, int M, int nacc, long n_out, mfb_shape *sh, size_t *lds_bytes, int planes = 3)
{
    sh->T = T;
    sh->M = M;
    sh->tpad = planes == 2 ? ((T - 1 + 7) & ~7) : ((T - 1 + 3) & ~3);   // int16 rows: 16-byte aligned at 8 samples
    sh->ksteps = (sh->tpad + 15 * M + 1 + 31) / 32;
    const int tile_out = MF_WAVES * nacc * 256;
    sh->total = ((tile_out - 16) * M + 32 * sh->ksteps + 7) & ~7;
    sh->plane = sh->total + 8;
    sh->tiles_per_ch = (int)((n_out + tile_out - 1) / tile_out);
    *lds_bytes = (size_t)(3 * sh->ksteps * 512 + planes * sh->plane) * 2;
    return *lds_bytes <= 160 * 1024 && sh->total <= MF_NV_MAX * MF_THREADS * 4;
}

int mfb_pick_nacc(int T, int M)
{
    mfb_shape sh;
    size_t bytes;
    if (const int v = llzs_tune(LLZS_TUNE_MFMA_NACC); (v == 1 || v == 2) && mfb_make_shape(T, M, v, 1, &sh, &bytes)) return v;
    if (mfb_make_shape(T, M, 2, 1, &sh, &bytes) && bytes <= 78 * 1024) return 2;
    if (mfb_make_shape(T, M, 1, 1, &sh, &bytes)) return 1;
    return 0;
}

template <int NACC, int NV>
int mfb_launch(const float *in, float *out, const float *hist, const float *taps, int channels, long n_in, long n_out,
               long in_pitch, long out_pitch, int T, int M, float gain, void *stream)
{
    mfb_shape sh;
    size_t lds_bytes;
    mfb_make_shape(T, M, NACC, n_out, &sh, &lds_bytes);
    if (lds_bytes > 64 * 1024)
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir_mfma_bf16x3<NACC, NV>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const long ntiles = (long)sh.tiles_per_ch * channels;
    int cus = 256, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    int per_cu = (int)((160 * 1024) / lds_bytes);
    if (per_cu > 4) per_cu = 4;
    if (const int v = llzs_tune(LLZS_TUNE_MFMA_WG_PER_CU); v >= 1 && v <= 8) per_cu = v;
    long grid = (long)cus * per_cu;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((k_fir_mfma_bf16x3<NACC, NV>), dim3((unsigned)grid), dim3(MF_THREADS), lds_bytes, as_stream(stream),
                       in, out, hist, taps, n_in, n_out, in_pitch, out_pitch, gain, sh, ntiles);
    LLZ_LAUNCH_CHECK("k_fir_mfma_bf16x3");
    return LLZ_OK;
}


template <int NACC, int NV>
int mfi_launch(const short *in, short *out, const short *hist, const float *taps, int channels, long n_in, long n_out,
               long in_pitch, long out_pitch, int T, int M, float gain, void *stream)
{
    mfb_shape sh;
    size_t lds_bytes;
    mfb_make_shape(T, M, NACC, n_out, &sh, &lds_bytes, 2);
    if (lds_bytes > 64 * 1024)
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir_mfma_i16<NACC, NV>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const long ntiles = (long)sh.tiles_per_ch * channels;
    int cus = 256, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    int per_cu = (int)((160 * 1024) / lds_bytes);
    if (per_cu > 4) per_cu = 4;
    long grid = (long)cus * per_cu;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((k_fir_mfma_i16<NACC, NV>), dim3((unsigned)grid), dim3(MF_THREADS), lds_bytes, as_stream(stream), in,
                       out, hist, taps, n_in, n_out, in_pitch, out_pitch, gain, sh, ntiles);
    LLZ_LAUNCH_CHECK("k_fir_mfma_i16");
    return LLZ_OK;
}

int mfi_pick_nacc(int T, int M)
{
    mfb_shape sh;
    size_t bytes;
    if (mfb_make_shape(T, M, 2, 1, &sh, &bytes, 2) && bytes <= 78 * 1024) return 2;
    if (mfb_make_shape(T, M, 1, 1, &sh, &bytes, 2)) return 1;
    return 0;
}

} // namespace

extern "C" int llzs_fir_mfma_f32_fits(int T, int M)
{
    return T >= 1 && M >= 1 && mfb_pick_nacc(T, M) > 0;
}

extern "C" int llzs_fir_mfma_f32(const float *in, float *out, const float *hist, const float *taps, int channels,
                                 long n_in, long n_out, long in_pitch, long out_pitch, int T, int M, float gain,
                                 void *stream)
{
    if (!in || !out || !taps || channels <= 0 || n_in <= 0 || n_out <= 0 || T < 1 || M < 1 ||
        in_pitch < n_in || out_pitch < n_out || (n_out - 1) * M >= n_in) {
        llzs_set_error("fir_mfma_f32: bad arguments (channels=%d n_in=%ld n_out=%ld T=%d M=%d)", channels, n_in, n_out,
                       T, M);
        return LLZ_ERR_ARG;
    }
    const int nb = mfb_pick_nacc(T, M);
    if (!nb) {
        llzs_set_error("fir_mfma_f32: %d taps at decimation %d do not fit the LDS image", T, M);
        return LLZ_ERR_RANGE;
    }
    mfb_shape sh;
    size_t bytes;
    mfb_make_shape(T, M, nb, n_out, &sh, &bytes);
    const bool small = sh.total <= 8 * MF_THREADS * 4;               // fewer prefetch registers
#define MFB_GO(A, V) return mfb_launch<A, V>(in, out, hist, taps, channels, n_in, n_out, in_pitch, out_pitch, T, M, gain, stream)
    if (nb == 2) { if (small) MFB_GO(2, 8); else MFB_GO(2, 16); }
    if (small) MFB_GO(1, 8); else MFB_GO(1, 16);
#undef MFB_GO
}

extern "C" int llzs_fir_mfma_i16_fits(int T, int M)
{
    return T >= 1 && M >= 1 && mfi_pick_nacc(T, M) > 0;
}

// int16 in / out, fp32 accumulate on the matrix cores: within 1 LSB of the reference, NOT bit-exact (see k_fir_mfma_i16)
extern "C" int llzs_fir_mfma_i16(const short *in, short *out, const short *hist, const float *taps, int channels,
                                 long n_in, long n_out, long in_pitch, long out_pitch, int T, int M, float gain,
                                 void *stream)
{
    if (!in || !out || !taps || channels <= 0 || n_in <= 0 || n_out <= 0 || T < 1 || M < 1 ||
        in_pitch < n_in || out_pitch < n_out || (n_out - 1) * M >= n_in) {
        llzs_set_error("fir_mfma_i16: bad arguments (channels=%d n_in=%ld n_out=%ld T=%d M=%d)", channels, n_in, n_out,
                       T, M);
        return LLZ_ERR_ARG;
    }
    const int nb = mfi_pick_nacc(T, M);
    if (!nb) {
        llzs_set_error("fir_mfma_i16: %d taps at decimation %d do not fit the LDS image", T, M);
        return LLZ_ERR_RANGE;
    }
    mfb_shape sh;
    size_t bytes;
    mfb_make_shape(T, M, nb, n_out, &sh, &bytes, 2);
    const bool small = sh.total <= 8 * MF_THREADS * 8;               // 8 samples per 16-byte prefetch register
#define MFI_GO(A, V) return mfi_launch<A, V>(in, out, hist, taps, channels, n_in, n_out, in_pitch, out_pitch, T, M, gain, stream)
    if (nb == 2) { if (small) MFI_GO(2, 8); else MFI_GO(2, 16); }
    if (small) MFI_GO(1, 8); else MFI_GO(1, 16);
#undef MFI_GO
}
