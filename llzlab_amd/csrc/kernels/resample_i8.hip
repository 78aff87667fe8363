// resample_i8.hip -- K3gx: the reference's int16 rational L/M resampler (llz_resample.c:583-603), BIT-EXACT, screened on the
// int8 matrix cores.  The general-L/M sibling of fir_mfma_i8.hip (L = 1), on the mapping of resample_mfma.hip.
//
//   reference, per output i = L m + f (period m, phase f):   y = 0;  for k = 0 .. Q-1:  y += (double)x[M m + c_f - k] * g[f][k];
//                                                            y *= gain;  clamp to [-32768, 32767];  (short)y        c_f = (f M) / L
//
// As in the L = 1 kernel the rounding order of that double loop only decides outputs whose value lies within ~1e-6 of an
// integer, so every output is first computed in exact INTEGER arithmetic -- taps quantised per phase to five balanced base-256
// digits, samples split into two byte planes, nine digit-plane products per 64 window samples on v_mfma_i32_16x16x64_i8, the
// 32-bit decision of screen_i8.hpp -- and only the undecided ones (about 2 eps of all) are recomputed in the reference's order
// by the lane that found them.
//
// Mapping.  For ONE period every phase reads inside the same window of the input, and the taps do not depend on the period:
// Y[f][m] = sum_u W[f][u] X[u][m] with a FIXED banded matrix.  A tile is 16 phases x 16 periods: D[r][n] = output (phase 16 t + r,
// period n).  The band of a phase tile starts at window position a_t = c_{16t} + Hp - (Q-1) (Hp = Q-1 rounded up to 8 = the
// position of a period's own first sample) and is walked 64 samples per step -- ONE step at 147:160 and 160:147, where 16
// phases drift by at most 17 samples and Q = 47; A (the tile's tap digits, [step][plane] in operand order, built on the host)
// stays in the wave's registers while it walks the period tiles of a span.  The LDS image is the span's samples as two
// CONTIGUOUS byte planes (low digit (x & 255) - 128, high digit x >> 8; written 8 samples at a time): period n's window starts
// M n bytes into a plane, so a B operand is ONE ds_read_b128 at an arbitrary byte address -- unaligned 16-byte LDS reads are
// legal on gfx950 (it is what hipcc itself emits for an under-aligned load; measured cost 13 %), and no second copy of any
// sample is needed (a first version kept one aligned column per period: 1.7 copies of every sample, byte stores when M is
// odd -- 3.6 ms against 1.2 ms at 160:147 / 147:160).
//
// Structure of the kernel (its first form -- output image in LDS, two barriers per span, tiles dealt over at most eight waves,
// per-tile constants fetched per span -- took 0.99 / 1.02 ms at 147:160 / 160:147; profiles/r03/ab_resample_i8x_general.txt has
// the phase times that led here: 0.57 / 0.68 ms):
//   * every wave owns ONE phase tile for the whole launch where the tiles fit the workgroup (twelve waves: L <= 192) and keeps
//     its digits, start values and masks in registers; more tiles are dealt evenly over the waves and set up per span;
//   * four consecutive phases of one period are 8 contiguous bytes of the output row: one global store per lane and period
//     tile at an even byte address -- no output image in LDS (its unaligned 8-byte LDS writes measured ~650 clocks per period
//     tile), no copy-out phase, no second barrier.  The waves of a period fill its 2 L bytes within one span, so the lines
//     leave L2 whole;
//   * the planes are DOUBLE-BUFFERED: span i + 1 is written while slower waves still read span i -- ONE barrier per span;
//   * the next span's samples are requested by hand (global_load ... from an SGPR base, as in fir_mfma_i8.hip) and awaited
//     with the count of younger stores, so staging does not wait for the previous span's stores to be acknowledged.  Spans at
//     a frame's edges (history in front, zeros behind) and the first span of a walk are staged sample by sample.
#include "screen_i8.hpp"
#include <math.h>

// RI_TRACE (a measurement build only, tools/trace_i16.py says how): shader-clock time of each phase of the span loop, summed per wave
#ifdef RI_TRACE
__device__ unsigned long long ri_trace_buf[65536 * 10];
#define RI_T0() unsigned long long ri_t = __builtin_readcyclecounter(), ri_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned ri_cnt[2] = {0, 0}
#define RI_MARK(i) do { const unsigned long long now = __builtin_readcyclecounter(); ri_acc[i] += now - ri_t; ri_t = now; } while (0)
#define RI_DUMP() do { if ((threadIdx.x & 63) == 0) { const unsigned w = ((blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 65535u; \
    for (int i = 0; i < 10; i++) ri_trace_buf[w * 10 + i] = ri_acc[i]; } \
    { unsigned total = 0; for (int l = 0; l < 64; l++) total += __shfl(ri_cnt[1], l); if ((threadIdx.x & 63) == 0) { const unsigned w = ((blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 65535u; ri_trace_buf[w * 10 + 3] = ((unsigned long long)ri_cnt[0] << 32) | total; } } } while (0)
#else
#define RI_T0() do { } while (0)
#define RI_MARK(i) do { } while (0)
#define RI_DUMP() do { } while (0)
#endif

namespace {

typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int RI_NG = 4;                    // 16-byte sample groups a thread holds for the next span (at most)

struct ri_shape {
    int L, M, Q, nt;        // nt = ceil(L / 16) phase tiles
    int Hp;                 // window position of the period's own first sample (u = 0): Q - 1 rounded up to 8
    int pt, P;              // period tiles per span, periods per span = 16 pt
    int plane;              // bytes per plane (multiple of 16): 8 ngroups
    int ngroups;            // 16-byte sample groups per span
    long spans;             // spans of a channel
    long periods;           // periods of a channel (n_out / L, rounded up)
    int spans_per_wg;
    int rs;                 // shift - 40
    unsigned e32;           // ceil(eps 2^32) + 2, eps = the largest phase's bound
    double gain;
};

struct __attribute__((packed, aligned(1))) ri_b128 { scr_i32x4 v; };      // a 16-byte LDS read at any byte address

// the reference's loop for one output from the planes: the sample at image position p is 256 hi[p] + lo[p] + 128
// (taps k < k0 and k > k1 are zero: adding x * 0 = +-0 to the running sum changes nothing, bit for bit -- phase 0 of an
//  interpolating ratio has ONE non-zero tap, and every output of it lands within 1e-12 of an integer)
__device__ __forceinline__ short ri_exact(const signed char *hi, const signed char *lo, int p0, const double *__restrict__ g, int k0,
                                          int k1, double gain)
{
#pragma clang fp contract(off)
    double y = 0.0;
#pragma unroll 1
    for (int k = k0; k <= k1; k++) {
        const int xv = 256 * (int)hi[p0 - k] + (int)lo[p0 - k] + 128;
        const double prod = (double)xv * g[k];
        y = y + prod;
    }
    y = y * gain;                                   // llz_resample.c:594
    if (y > 32767) y = 32767;
    if (y < -32768) y = -32768;
    return (short)y;                                // :601, toward zero
}

struct __attribute__((packed, aligned(2))) ri_g64 { scr_i16x4 v; };        // an 8-byte global store at any even byte address

__device__ __forceinline__ i16x8 ri_load_nt(const short *base, int off)
{
    i16x8 r;
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(r) : "v"(off), "s"(base) : "memory");
    return r;
}
// at most n vector-memory operations (the youngest) still in flight; n = 0 .. 8
__device__ __forceinline__ void ri_wait_vm(int n)
{
    switch (n) {
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void ri_pin(i16x8 (&v)[RI_NG])
{
    static_assert(RI_NG == 4, "one operand per group");
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory");
}

constexpr int RI_D_THREADS = 768;           // twelve waves: 170 VGPRs a lane

// EXACTS: some phase is exact (per-lane bit-field widths in the decision: four more registers); MULTI: more phase tiles than waves
template <int KS, bool NEG, bool EXACTS, bool MULTI>
__global__ void __launch_bounds__(KS <= 2 ? RI_D_THREADS : 512)
k_resample_i8d(const short *__restrict__ in, short *__restrict__ out, const short *__restrict__ hist,
               const signed char *__restrict__ atab, const int *__restrict__ aoff, const int *__restrict__ bqtab,
               const double *__restrict__ g, long n_in, long n_out, long in_pitch, long out_pitch, ri_shape sh)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // planes of buffer b: low digits at 2 b plane, high digits one plane further
    const int tid = threadIdx.x, lane = tid & 63, threads = (int)blockDim.x, waves = threads >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int col16 = scr_col(n), ck = scr_chunk(kq);
    const int c = blockIdx.y;
    const short *row = in + (size_t)c * in_pitch;
    const short *hrow = hist ? hist + (size_t)c * (sh.Q - 1) : nullptr;
    short *orow = out + (size_t)c * out_pitch;
    const bool aligned_in = (in_pitch & 7) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const int P = sh.P;

    // the lane's groups of a span (8 samples = 16 bytes each): byte offset of group k; lanes past the last group re-read it and
    // write nothing
    auto goff = [&](int k) {
        const int gi = k * threads + tid;
        return 16 * (gi < sh.ngroups ? gi : sh.ngroups - 1);
    };
    // first sample index of a span's image: the 8-aligned index at or below M m0 - Hp
    auto image_base = [&](long sp, int &lead) {
        const long s0 = sp * P * sh.M - sh.Hp;
        const long gbase = s0 >= 0 ? (s0 & ~7L) : -((-s0 + 7) & ~7L);
        lead = (int)(s0 - gbase);
        return gbase;
    };
    auto streams = [&](long gbase) { return aligned_in && gbase >= 0 && gbase + 8L * sh.ngroups <= n_in; };

    // A tile's state: digits, band start, start values of the lane's four phases, and the slots that never need a second look
    // (lane masks): the EXACT phases (a single tap 1.0, or no tap: the outputs are integers -- always "unsure", and not to be
    // moved toward zero: bit-field width 0 in the decision) and the rows past the last phase.  MULTI = false: the wave owns tile
    // `wave` for the whole launch and sets this up once; MULTI = true (more phase tiles than waves): tiles wave, wave + waves,
    // ... per span, set up again for each (global loads that hit L2).
    int t = wave, f0 = 16 * wave + 4 * kq;
    bool tile_whole = false;                                                     // every row of the tile is a phase
    scr_i32x4 ad[KS][5];
    int a_t = 0;
    scr_i32x4 start[3] = {};
    unsigned long long settled[4] = {0, 0, 0, 0};
    unsigned inexact[4] = {1u, 1u, 1u, 1u};
    unsigned e32v = sh.e32;                             // (in a VGPR: an instruction takes one scalar operand, and the shift is one)
    auto tile_setup = [&](int tile) {
        t = tile;
        f0 = 16 * t + 4 * kq;
        tile_whole = 16 * t + 15 < sh.L;
        const signed char *ap = atab + ((size_t)t * KS * 5) * 1024 + lane * 16;
#pragma unroll
        for (int s = 0; s < KS; s++)
#pragma unroll
            for (int p = 0; p < 5; p++) ad[s][p] = *reinterpret_cast<const scr_i32x4 *>(ap + (s * 5 + p) * 1024);
        a_t = aoff[t];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const scr_i32x4 row4 = *reinterpret_cast<const scr_i32x4 *>(bqtab + 4 * (f0 + j));
            int s0, s2, s4;
            scr_start(row4[0], row4[1], s0, s2, s4);
            start[0][j] = s0;
            start[1][j] = s2;
            start[2][j] = s4;
            inexact[j] = (row4[3] >> 16) != 0 ? 0u : 1u;
            settled[j] = __ballot(inexact[j] == 0 || f0 + j >= sh.L);
        }
    };
    const bool has_tile = wave < sh.nt;
    if (!MULTI) {
        if (has_tile) {
            tile_setup(wave);
        } else {
#pragma unroll
            for (int s = 0; s < KS; s++)
#pragma unroll
                for (int p = 0; p < 5; p++) ad[s][p] = (scr_i32x4){0, 0, 0, 0};
        }
        // The tile's registers pass through one empty asm HERE: the compiler waits for their loads in front of it, once.  Left
        // to itself it keeps them "possibly in flight" at the head of the period-tile loop and puts s_waitcnt vmcnt(0) in front
        // of the first products of EVERY period tile -- which also waits for the previous tile's store to be acknowledged.
#pragma unroll
        for (int s = 0; s < KS; s++)
            asm volatile("" : "+v"(ad[s][0]), "+v"(ad[s][1]), "+v"(ad[s][2]), "+v"(ad[s][3]), "+v"(ad[s][4]) : : "memory");
        asm volatile("" : "+v"(start[0]), "+v"(start[1]), "+v"(start[2]), "+v"(e32v) : : "memory");
        if constexpr (EXACTS) asm volatile("" : "+v"(inexact[0]), "+v"(inexact[1]), "+v"(inexact[2]), "+v"(inexact[3]) : : "memory");
    }

    RI_T0();
    // ---- a span's samples into the planes of buffer b ----
    auto stage = [&](long sp, int b, i16x8 (&v)[RI_NG], bool streamed, int young) {
        signed char *lo_p = reinterpret_cast<signed char *>(lds) + 2 * b * sh.plane, *hi_p = lo_p + sh.plane;
        if (streamed) {
            ri_wait_vm(young);                          // the request is older than exactly `young` stores (0: do not count)
            ri_pin(v);
#pragma unroll
            for (int k = 0; k < RI_NG; k++) {
                const int gi = k * threads + tid;
                if (gi < sh.ngroups) {
                    const u32x4 d = __builtin_bit_cast(u32x4, v[k]);
                    u32x2 lo, hi;
                    lo[0] = __builtin_amdgcn_perm(d[1], d[0], 0x06040200u) ^ 0x80808080u;
                    lo[1] = __builtin_amdgcn_perm(d[3], d[2], 0x06040200u) ^ 0x80808080u;
                    hi[0] = __builtin_amdgcn_perm(d[1], d[0], 0x07050301u);
                    hi[1] = __builtin_amdgcn_perm(d[3], d[2], 0x07050301u);
                    *reinterpret_cast<u32x2 *>(&lo_p[8 * gi]) = lo;
                    *reinterpret_cast<u32x2 *>(&hi_p[8 * gi]) = hi;
                }
            }
        } else {
            // a frame edge (history in front, zeros behind), an unaligned frame, or the first span of the walk
            int lead;
            const long gbase = image_base(sp, lead);
            for (int e = tid; e < 8 * sh.ngroups; e += threads) {
                const long idx = gbase + e;
                int x = 0;
                if (idx >= 0) {
                    if (idx < n_in) x = row[idx];
                } else if (hrow && idx >= -(long)(sh.Q - 1)) {
                    x = hrow[sh.Q - 1 + idx];
                }
                lo_p[e] = (signed char)((x & 255) - 128);
                hi_p[e] = (signed char)(x >> 8);
            }
        }
    };

    // ---- products, decisions and stores of the wave's tile over the span in buffer b; returns the number of store
    //      instructions issued (0 when it is not the same for every lane).  The loop is bound by the NUMBER of instructions a
    //      wave issues (the waves of a SIMD leave the barrier together and take turns, one instruction each): operand and
    //      output addresses advance by increments, the store takes an SGPR base and a 32-bit lane offset, the exact phases are
    //      a per-lane bit-field width inside the decision instead of a second pass. ----
    auto products = [&](long sp, int b) {
        const signed char *lo_p = reinterpret_cast<const signed char *>(lds) + 2 * b * sh.plane;
        int lead;
        (void)image_base(sp, lead);
        const long m0 = sp * P;
        const long left = sh.periods - m0;                                        // periods of this span that exist
        const bool whole = tile_whole && left >= P;
        short *ospan = orow + m0 * sh.L;                                          // (wave-uniform)
        unsigned b_at = (unsigned)(2 * b * sh.plane + lead + col16 * sh.M + a_t + 16 * ck);       // LDS byte address of the lane's B operand
        unsigned o_at = 2u * (unsigned)(col16 * sh.L + f0);                       // byte offset of the lane's 8 output bytes in the span
        const unsigned b_step = 16u * (unsigned)sh.M, o_step = 32u * (unsigned)sh.L;
        const int tail = f0 + 3 < sh.L ? 4 : (f0 < sh.L ? sh.L - f0 : 0);         // values of the lane that are phases
#pragma unroll 1
        for (int p = 0; p < sh.pt; p++, b_at += b_step, o_at += o_step) {
            const signed char *bp = reinterpret_cast<const signed char *>(lds) + b_at;
            scr_i32x4 acc[5];
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const scr_i32x4 b_lo = reinterpret_cast<const ri_b128 *>(bp + 64 * s)->v;
                const scr_i32x4 b_hi = reinterpret_cast<const ri_b128 *>(bp + sh.plane + 64 * s)->v;
                if (s == 0) scr_step<true>(acc, ad[s], b_lo, b_hi, start);
                else scr_step<false>(acc, ad[s], b_lo, b_hi, start);
            }
            int res[4];
            bool unsure[4];
            unsigned long long open = 0;                                          // lanes with an undecided slot (a scalar mask)
#ifdef RI_TRACE
            asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[4][0]));             // (the products have landed)
#endif
            RI_MARK(7);                                 // operand reads and products
#pragma unroll
            for (int j = 0; j < 4; j++) {
                res[j] = scr_decide<NEG>(acc[0][j], acc[1][j], acc[2][j], acc[3][j], acc[4][j], sh.rs, e32v, unsure[j], EXACTS ? inexact[j] : 1u,
                                         2u * sh.e32);
                open |= __ballot(unsure[j]) & ~settled[j];
            }
            if (open != 0) {
                const int col = 16 * p + col16;
                unsigned mine = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    // (not the settled slots, and not a value within eps of zero: that truncates to 0 from either side)
                    const bool look = unsure[j] && !((settled[j] >> lane) & 1ull) &&
                                      !scr_near_zero<NEG>(acc[0][j], acc[1][j], acc[2][j], acc[3][j], acc[4][j], sh.rs, sh.e32);
                    mine = mine + mine + (look ? 1u : 0u);                                 // slot j at bit 3 - j
                }
#pragma unroll 1
                while (mine != 0) {
                    // an integer within eps of the value: the reference's own arithmetic decides, in the lane that found it
                    const int j = 3 - __builtin_ctz(mine);
                    mine &= mine - 1;
                    const int f = f0 + j;
                    int cur = 0;
#pragma unroll
                    for (int u = 0; u < 4; u++) cur = j == u ? res[u] : cur;
                    // (a value far outside the clamp range needs no second look, nor does a period that does not exist)
                    if (col < left && scr_in_reach(cur)) {
                        const int cf = (int)(((long)f * sh.M) / sh.L);
                        const int kr = bqtab[4 * f + 3] & 0xffff;                 // first | last << 8 non-zero tap of the phase
                        const int r = ri_exact(lo_p + sh.plane, lo_p, lead + col * sh.M + cf + sh.Hp, g + (size_t)f * sh.Q, kr & 255,
                                               kr >> 8, sh.gain);
#pragma unroll
                        for (int u = 0; u < 4; u++) res[u] = j == u ? r : res[u];
                    }
                }
            }
            // four consecutive phases of one period: 8 bytes of the output row, at any even byte address
            const scr_i16x4 y = scr_clamp4(res);
#ifdef RI_TRACE
            asm volatile("s_nop 0" ::"v"(y), "s"(open));
#endif
            RI_MARK(8);                                 // decisions
            if (whole) {
                asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(o_at), "v"(y), "s"(ospan) : "memory");
            } else if (16 * p + col16 < left) {
                // the ragged last tile (and the last span): whole lanes as above; the lane group that straddles the last phase
                // stores its 1..3 values as a 4-byte and / or a 2-byte piece (tail: the same for the whole launch)
                if (tail == 4) {
                    asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(o_at), "v"(y), "s"(ospan) : "memory");
                } else if (tail > 0) {
                    const scr_i16x2 y01 = {y[0], y[1]};
                    if (tail >= 2) asm volatile("global_store_dword %0, %1, %2" : : "v"(o_at), "v"(y01), "s"(ospan) : "memory");
                    if (tail == 1) asm volatile("global_store_short %0, %1, %2" : : "v"(o_at), "v"(y01), "s"(ospan) : "memory");
                    if (tail == 3) {
                        const int y2 = y[2];
                        asm volatile("global_store_short %0, %1, %2 offset:4" : : "v"(o_at), "v"(y2), "s"(ospan) : "memory");
                    }
                }
            }
            RI_MARK(9);                                 // stores
        }
        return whole ? sh.pt : 0;
    };

    // One loop, one request site.  Iteration i runs the products of span i (none in the first iteration), writes span i + 1
    // into the other planes and requests span i + 2; the barrier at its end publishes the planes of i + 1 and retires the
    // readers of i.  A request that does not stream (or has no span) reads the row's first groups instead: in bounds -- the
    // launcher requires n_in >= 8 ngroups -- and never looked at.
    const long span0 = (long)blockIdx.x * sh.spans_per_wg;
    const long span1 = min(span0 + sh.spans_per_wg, sh.spans);
    i16x8 v[RI_NG] = {};
    bool requested = false;                             // the registers hold span i + 1
    int b = 0;
    for (long i = span0 - 1; i < span1; i++) {
        int young = 0;
        if (i >= span0) {
            if (!MULTI) {
                if (has_tile) young = products(i, b);
            } else {
                bool counted = true;
                for (int tile = wave; tile < sh.nt; tile += waves) {
                    tile_setup(tile);
                    const int stores = products(i, b);
                    counted = counted && stores > 0;
                    young += stores;
                }
                // (fewer than the true number of younger stores only waits for more; the tile loads above are younger too)
                young = counted ? (young < 8 ? young : 8) : 0;
            }
        }
        RI_MARK(6);                                     // (the rest of the products phase)
        if (i + 1 < span1) stage(i + 1, b ^ 1, v, requested, young);
        RI_MARK(0);                                     // planes written
        int lead;
        const long gnext = image_base(i + 2, lead);
        requested = i + 2 < span1 && streams(gnext);
        const short *src = requested ? row + gnext : row;
#pragma unroll
        for (int k = 0; k < RI_NG; k++) v[k] = ri_load_nt(src, goff(k));
        RI_MARK(1);                                     // next span requested
        __syncthreads();
        RI_MARK(2);                                     // the barrier
        b ^= 1;
    }
    RI_DUMP();
}

// steps of 64 window samples the widest band needs: the band of tile t spans positions a_t .. c_{last phase} + Hp
int ri_ksteps(int L, int M, int Q)
{
    const int nt = (L + 15) / 16;
    int ks = 1;
    for (int t = 0; t < nt; t++) {
        const int flast = 16 * t + 15 < L ? 16 * t + 15 : L - 1;
        const int width = (int)(((long)flast * M) / L - ((long)16 * t * M) / L) + Q;
        const int need = (width + 63) / 64;
        if (need > ks) ks = need;
    }
    return ks;
}

} // namespace

extern "C" int llzs_resample_i16x_ksteps(int L, int M, int Q) { return ri_ksteps(L, M, Q); }

// geometry of the launch (without the walk length: ri_pick_walk): a wave per phase tile where the tiles fit the workgroup (twelve
// waves at up to two steps, eight above), else the tiles dealt evenly over the waves; up to 8 period tiles per span in two pairs
// of planes of at most 64 KB together; n_in > 0: no span longer than the frame (requests without a span read the row's head)
static bool ri_make_shape(int L, int M, int Q, long n_in, long n_out, ri_shape *sh, int *waves, size_t *lds)
{
    sh->L = L; sh->M = M; sh->Q = Q;
    sh->nt = (L + 15) / 16;
    sh->Hp = (Q - 1 + 7) & ~7;
    const int ks = ri_ksteps(L, M, Q);
    if (ks > 4) return false;
    const int wmax = ks <= 2 ? RI_D_THREADS / 64 : 8;
    const int rounds = (sh->nt + wmax - 1) / wmax;
    int w = (sh->nt + rounds - 1) / rounds;
    if (w < 2) w = 2;
    const long periods = (n_out + L - 1) / L;
    bool ok = false;
    const int pt_forced = llzs_tune(LLZS_TUNE_RS_I16_TILES);
    for (int pt = (pt_forced >= 1 && pt_forced <= 8) ? pt_forced : 8; pt >= 1 && !ok; pt--) {
        if (pt > 1 && 16L * (pt - 1) >= periods) continue;              // (no span longer than the signal needs)
        sh->pt = pt; sh->P = 16 * pt;
        // the last period's last band ends at most M - 1 + Hp + 64 ks positions into its window; + 7 of alignment slack in front
        const long bytes = 7 + (long)M * (sh->P - 1) + (M - 1) + sh->Hp + 64 * ks + 16;
        sh->ngroups = (int)((bytes + 7) / 8);
        sh->plane = (8 * sh->ngroups + 15) & ~15;
        *lds = 4 * (size_t)sh->plane;
        *waves = w;
        if (pt == 1)
            while (*waves < wmax && sh->ngroups > RI_NG * 64 * *waves) (*waves)++;      // (a single tile may add staging waves)
        ok = sh->ngroups <= RI_NG * 64 * *waves && *lds <= 64 * 1024 && (n_in <= 0 || n_in >= 8L * sh->ngroups);
    }
    if (!ok) return false;
    sh->periods = periods;
    sh->spans = (periods + sh->P - 1) / sh->P;
    sh->spans_per_wg = 1;
    return true;
}

// consecutive spans per workgroup: the walk length that minimises rounds x (length + 1) when `resident` workgroups run at a time
// (every walk pays about one span of fill: its first request has nothing to hide behind)
static void ri_pick_walk(ri_shape *sh, int channels, long resident)
{
    long spw = sh->spans < 4 ? sh->spans : 4;
    double best = 1e300;
    for (long cnt = spw; cnt <= sh->spans; cnt++) {
        const long wgs = ((sh->spans + cnt - 1) / cnt) * (long)channels;
        const double cost = (double)((wgs + resident - 1) / resident) * (double)(cnt + 1);
        if (cost < best * 0.999) { best = cost; spw = cnt; }
    }
    const long forced = llzs_tune(LLZS_TUNE_RS_I16_WALK);
    if (forced >= 1) spw = forced < sh->spans ? forced : sh->spans;
    sh->spans_per_wg = (int)spw;
}

// the kernel instance of a shape, the workgroups of it a CU holds at a time (registers, LDS and waves of THAT instance), and the
// walk length that follows
static int ri_plan(bool exacts, int ks, bool neg, bool multi, int waves, size_t lds, int channels, ri_shape *sh, const void **fn,
                   int *per_cu)
{
#define RI_F4(K, N, E, U) reinterpret_cast<const void *>(k_resample_i8d<K, N, E, U>)
#define RI_F3(K, N, E) (multi ? RI_F4(K, N, E, true) : RI_F4(K, N, E, false))
#define RI_F2(K, N) (exacts ? RI_F3(K, N, true) : RI_F3(K, N, false))
#define RI_F(K) (neg ? RI_F2(K, true) : RI_F2(K, false))
    // (more than one step: the exact-phase registers are always there -- half the instances, and no register is short)
    *fn = ks == 1 ? RI_F(1) : neg ? (ks == 2 ? RI_F3(2, true, true) : ks == 3 ? RI_F3(3, true, true) : RI_F3(4, true, true))
                                  : (ks == 2 ? RI_F3(2, false, true) : ks == 3 ? RI_F3(3, false, true) : RI_F3(4, false, true));
#undef RI_F
#undef RI_F2
#undef RI_F3
#undef RI_F4
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus < 1) {
        llzs_set_error("resample_i16x: cannot query the device");
        return LLZ_ERR_DEVICE;
    }
    if (lds > 64 * 1024) LLZ_HIP_CHECK(hipFuncSetAttribute(*fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, *fn, 64 * waves, lds) != hipSuccess || *per_cu < 1) *per_cu = 1;
    ri_pick_walk(sh, channels, (long)*per_cu * cus);
    return LLZ_OK;
}

// the launch a call would make (measurement and documentation): plan[0..6] = waves per workgroup, periods per span, spans per
// workgroup, workgroups, workgroups resident per CU, LDS bytes per workgroup, 1 when a wave takes several phase tiles
extern "C" int llzs_resample_i16x_plan(int L, int M, int Q, int channels, long n_out, int shift, int *plan)
{
    ri_shape sh;
    int waves, per_cu = 0;
    size_t lds;
    const void *fn = nullptr;
    if (!plan || channels < 1 || n_out < 1 || !ri_make_shape(L, M, Q, (n_out / L) * M, n_out, &sh, &waves, &lds)) {
        llzs_set_error("resample_i16x_plan: bad arguments");
        return LLZ_ERR_ARG;
    }
    const int rc = ri_plan(false, ri_ksteps(L, M, Q), shift - 40 < 0, sh.nt > waves, waves, lds, channels, &sh, &fn, &per_cu);
    if (rc != LLZ_OK) return rc;
    plan[0] = waves; plan[1] = sh.P; plan[2] = sh.spans_per_wg;
    plan[3] = (int)((sh.spans + sh.spans_per_wg - 1) / sh.spans_per_wg) * channels;
    plan[4] = per_cu; plan[5] = (int)lds; plan[6] = sh.nt > waves ? 1 : 0;
    return LLZ_OK;
}

extern "C" int llzs_resample_i16x_fits(int L, int M, int Q)
{
    if (L < 2 || M < 1 || Q < 1 || Q > 200) return 0;
    ri_shape sh;
    int waves;
    size_t lds;
    return ri_make_shape(L, M, Q, 0, L, &sh, &waves, &lds) ? 1 : 0;
}

// atab: [ceil(L/16)][ksteps][5][64][16] tap digits in operand order; aoff: [ceil(L/16)] band starts; bqtab: [16 ceil(L/16)][4]
// per phase: floor(128 sum_k G_f[k] / 256) as (lo, hi), e32 = ceil(eps_f 2^32) + 2, and first | last << 8 non-zero tap | flag << 16
// with flag = 1 when the phase's outputs are exact integers the screen itself reproduces (a single tap 1.0 at gain 1.0: no
// second look); g: the L x Q double taps; shift as
// in llzs_fir_mfma_i16x; eps = the largest per-phase bound (checked only).  The call must
// start on a period boundary (input index % M == 0, output index % L == 0).
extern "C" int llzs_resample_i16x(const short *in, short *out, const short *hist, const signed char *atab, const int *aoff,
                                  const int *bqtab, const double *g, int channels, long n_in, long n_out, long in_pitch,
                                  long out_pitch, int L, int M, int Q, int shift, double gain, double eps, int any_exact, void *stream)
{
    ri_shape sh;
    int waves;
    size_t lds;
    if (!in || !out || !atab || !aoff || !bqtab || !g || channels <= 0 || channels > 65535 || n_in <= 0 || n_out <= 0 ||
        in_pitch < n_in || out_pitch < n_out || shift < 32 || shift > 46 || !(eps > 0.0) || !(eps < 0.0625) ||
        !llzs_resample_i16x_fits(L, M, Q)) {
        llzs_set_error("resample_i16x: bad arguments (channels=%d L=%d M=%d Q=%d shift=%d eps=%g)", channels, L, M, Q, shift, eps);
        return LLZ_ERR_ARG;
    }
    // the kernel's sample requests are unconditional: one that has no span reads the head of the row instead, so a frame must be
    // at least one span's image long (shorter: the caller takes the all-double kernel)
    if (!ri_make_shape(L, M, Q, n_in, n_out, &sh, &waves, &lds)) {
        llzs_set_error("resample_i16x: a frame of %ld samples is shorter than one span's image at %d:%d", n_in, L, M);
        return LLZ_ERR_RANGE;
    }
    sh.rs = shift - 40;
    sh.e32 = (unsigned)ceil(ldexp(eps, 32)) + 2u;
    sh.gain = gain;
    const int ks = ri_ksteps(L, M, Q);
    const bool neg = sh.rs < 0, multi = sh.nt > waves;
    int per_cu = 0;
    const void *fn = nullptr;
    const int prc = ri_plan(any_exact != 0, ks, neg, multi, waves, lds, channels, &sh, &fn, &per_cu);
    if (prc != LLZ_OK) return prc;
    const dim3 grid((unsigned)((sh.spans + sh.spans_per_wg - 1) / sh.spans_per_wg), (unsigned)channels), block(64 * waves);
    void *args[] = {&in, &out, &hist, &atab, &aoff, &bqtab, &g, &n_in, &n_out, &in_pitch, &out_pitch, &sh};
    LLZ_HIP_CHECK(hipLaunchKernel(fn, grid, block, args, lds, as_stream(stream)));
    LLZ_LAUNCH_CHECK("k_resample_i8d");
    return LLZ_OK;
}

#ifdef RI_TRACE
extern "C" int llzs_ri_trace_read(unsigned long long *dst, int count)
{
    LLZ_HIP_CHECK(hipDeviceSynchronize());
    LLZ_HIP_CHECK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(ri_trace_buf), sizeof(unsigned long long) * (size_t)count));
    return LLZ_OK;
}
#endif
