// fir_mfma_i8.hip -- K3x: the reference's int16 decimating resampler (llz_resample.c:583-603 with L = 1), BIT-EXACT, at
// matrix-core speed.
//
//   reference, per output i:   y = 0;  for k = 0 .. Q-1:  y += (double)x[iM - k] * g[k];   y *= gain;
//                              clamp to [-32768, 32767];  (short)y   (truncation toward zero)
//
// Computing that sum as the reference does -- Q dependent, separately rounded double multiply-adds per output -- runs on
// the unfused FP64 rate: k_resample_i16_exact (resample.hip) takes 167 ms for 8192 ch x 4 Mi at 1:3, 7 % of the HBM
// roofline.  But the rounding order only matters for an output whose value lies within ~1e-9 of an integer; everywhere
// else ANY sufficiently accurate sum truncates to the same int16.  So:
//
//   1. SCREEN (matrix cores, exact integer arithmetic).  Taps are quantised once on the host to G[k] = round(g[k] 2^s),
//      |G| < 2^39, and written as five balanced base-256 digits G = sum_p d_p 256^p, d_p in [-128, 127]; a sample is
//      x = 256 xh + xl' + 128 with xh = x >> 8 and xl' = (x & 255) - 128, both int8.  v_mfma_i32_16x16x64_i8 accumulates
//      the digit-plane products in int32 -- exactly: |sum| <= 2 Q 2^14 -- into five accumulators by weight w = 1 .. 5 (nine
//      products; the tenth, low sample digit x lowest tap digit at weight 0, is at most 128 sum|d_0| and is left inside eps):
//            S = sum_w 256^w acc_w + 128 sum_k G[k]        (exact integers)     = sum_k x[iM - k] G[k] - (weight-0 product)
//      (the gain is folded into G: G[k] = round(gain g[k] 2^s)), so v = S 2^-s differs from the reference's y only by the
//      tap quantisation and the reference's own rounding:
//            |v - y| <= eps := 2 (32768 sum_k |gain g[k] - G[k] 2^-s|  +  |gain| (Q + 2) 2^-52 32768 sum_k |g[k]|) + 2^-30
//      (the host evaluates this for the handle's taps: ~4e-6 at Q = 134; DESIGN.md has the derivation).
//   2. DECIDE, in 32-bit integer arithmetic with explicit carries (mx_decide: ~18 vector instructions per output, no
//      branch): the integer part I and the top 32 bits F of the fraction of v; if F is farther than eps from both ends no
//      integer lies within eps of v, so v and y truncate (and clamp) to the same int16: I, plus one if negative.
//   3. RECOMPUTE the others -- about 2 eps of all outputs, ~1e-5 -- in the reference's exact order, in double, by the lane
//      that found them, from the tile's samples in LDS (rounded multiply, rounded add, ascending k: the loop above).
//      All-zero tiles (digital silence: every output would sit ON the integer 0) are written as zeros without arithmetic.
//
// The result is the reference's int16 for every sample by construction; the tests compare it bit for bit with the oracle on
// random, clipping, silent and DC inputs.  Mapping and data movement are fir_mfma.hip's (banded-Toeplitz product: a wave
// owns 16 consecutive 16-output segments of a channel, D[m][n] = sum_t A[m][t] B[t][n] with A[m][t] = G[mM + tpad - t],
// B[t][n] = sample t of segment n's window; persistent workgroups walk tiles, the next tile's samples prefetched into
// registers), with 64 window samples per MFMA step and one-byte planes in LDS.
#include "screen_i8.hpp"
#include <math.h>
#include <type_traits>
#include <stdlib.h>
#include <string.h>

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int MX_WAVES = 4;
constexpr int MX_THREADS = MX_WAVES * 64;
constexpr int MX_PLANES = 5;                // tap digit planes
constexpr int MX_ACCS = 5;                  // accumulators by weight: 256^1 .. 256^5 (the weight-0 product is not formed)
constexpr int MX_NV_MAX = 8;                // prefetch registers: 8 x 16 B (8 samples each) per thread

struct mx_shape {
    int T, M;
    int tpad;        // multiple of 8 >= T-1 (int16 rows stay 16-byte aligned at tile starts)
    int ksteps;      // MFMA steps of 64 window samples
    int total;       // samples staged per tile (multiple of 8)
    int plane;       // bytes per LDS sample plane (multiple of 16)
    int tiles_per_ch;
};

// The screen's value of an output is v = S 2^-shift with S = sum_k x[k] G[k].  The kernel forms
//     T' = sum_{w=1..5} 256^(w-1) acc_w + floor(bias / 256)          (S = 256 T' + (bias mod 256) + the weight-0 product;
// both leftovers are bounded by the host and sit inside eps), so v = T' 2^-(shift-8) up to eps.
struct mx_params {
    int bq_lo, bq_hi;        // floor(bias / 256) as a 64-bit two's complement pair; bias = 128 sum_k G[k]
    unsigned e32;            // ceil(eps 2^32) + 2: the screen's uncertainty in units of 2^-32 of an output step
    int rs;                  // shift - 40 in [-8, 6]: T' 2^-(32 + rs) is the output value
    double gain;             // only the recompute path multiplies by it, as the reference does
};

// the reference's loop for one output, from the tile's LDS image: sample p of the window is 256 hi[p] + lo[p] + 128; gd = the
// double taps, in LDS too (no vector-memory operation and no call in here: the compiler's wait counts for the sample
// prefetch and the output stores survive this rarely taken path)
__device__ __forceinline__ short mx_exact(const signed char *hi, const signed char *lo, int p0, const double *gd, int T,
                                          double gain)
{
#pragma clang fp contract(off)
    double y = 0.0;
#pragma unroll 1
    for (int k = 0; k < T; k++) {
        const int xv = 256 * (int)hi[p0 - k] + (int)lo[p0 - k] + 128;
        const double prod = (double)xv * gd[k];
        y = y + prod;
    }
    y = y * gain;                                   // llz_resample.c:594
    if (y > 32767) y = 32767;
    if (y < -32768) y = -32768;
    return (short)y;                                // :601, toward zero
}

// The sample prefetch is issued and awaited by hand.  Left to the compiler, every use of a prefetched register was preceded
// by s_waitcnt vmcnt(0): that also waits for the previous tile's output STORES, once per tile.  Vector-memory operations
// complete in issue order per wave, so a request issued `young` operations ago is complete once at most `young` operations
// are outstanding; the kernel counts the operations it issues behind a request (a lower bound is safe: it only makes the wait
// stricter).  Rules that keep this safe against the compiler, which does not know that the registers are pending:
//   * ONE request site in the whole kernel, unconditional (a tile that cannot use the streamed form requests a harmless
//     in-bounds address and ignores the data): a value defined in two places, or under a branch, is merged by copies -- and a
//     copy of a pending register reads (and may write back) stale data;
//   * ONE point behind the waits through which the registers pass (mx_pin, no instruction): every use depends on it, so
//     none can be placed in front of a wait.
// base: wave-uniform 64-bit address (SGPR pair); off: per-lane byte offset
__device__ __forceinline__ i16x8 mx_load_nt(const short *base, int off)
{
    i16x8 r;
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(r) : "v"(off), "s"(base) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void mx_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
template <int NV>
__device__ __forceinline__ void mx_pin(i16x8 (&v)[NV])
{
    if constexpr (NV == 4)
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory");
    else
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
}

__device__ __forceinline__ int mx_seg(int n) { return scr_col(n); }          // (screen_i8.hpp: why this order)
__host__ __device__ __forceinline__ int mx_chunk(int kq) { return scr_chunk(kq); }

} // namespace
// MX_TRACE (a measurement build only): shader-clock time of each phase of the tile loop, summed per wave (tools/trace_i16.py)
#ifdef MX_TRACE
__device__ unsigned long long mx_trace_buf[65536 * 8];
#define MX_T0() unsigned long long mx_t = __builtin_readcyclecounter(), mx_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define MX_MARK(i) do { const unsigned long long now = __builtin_readcyclecounter(); mx_acc[i] += now - mx_t; mx_t = now; } while (0)
#define MX_DUMP() do { if ((threadIdx.x & 63) == 0) { const unsigned w = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 65535u; \
    for (int i = 0; i < 8; i++) mx_trace_buf[w * 8 + i] = mx_acc[i]; } } while (0)
#else
#define MX_T0() do { } while (0)
#define MX_MARK(i) do { } while (0)
#define MX_DUMP() do { } while (0)
#endif
namespace {

// KS: the number of 64-sample steps when it is 1 or 2 (every LDS operand address is then base + constant), 0 = read sh.ksteps
// (three unrolled steps measured SLOWER than the loop on config 5: 130 registers against 123, 3 waves per SIMD against 4 --
// 21.9 against 21.6 ms, same box)
template <int NACC, int NV, bool NEG, int KS>
__global__ void __launch_bounds__(MX_THREADS)
k_fir_mfma_i8x(const short *__restrict__ in, short *__restrict__ out, const short *__restrict__ hist,
               const signed char *__restrict__ digits /* [MX_PLANES][T] */, const double *__restrict__ gd, long n_in,
               long n_out, long in_pitch, long out_pitch, mx_shape sh, mx_params pr, int channels)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int TILE_OUT = MX_WAVES * NACC * 256;
    const int ksteps = KS > 0 ? KS : sh.ksteps;
    const int abytes = MX_PLANES * ksteps * 1024;        // A table: [ksteps][planes][64 lanes][16 bytes]
    signed char *atab = reinterpret_cast<signed char *>(lds);
    signed char *xs_lo = atab + abytes;                  // sample planes: low digit (x & 255) - 128, then x >> 8
    signed char *xs_hi = xs_lo + sh.plane;
    double *gd_lds = reinterpret_cast<double *>(xs_hi + sh.plane);      // the T double taps (recompute path)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int seg = mx_seg(n);
    const int total = sh.total, last8 = sh.total - 8;
    const bool aligned_in = (in_pitch & 7) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const bool aligned_out = (out_pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0;

    for (int e = tid; e < abytes; e += MX_THREADS) {
        const int sp = e >> 10, s = sp / MX_PLANES, p = sp - s * MX_PLANES;
        const int l = (e >> 4) & 63, j = e & 15;
        const int t = 64 * s + 16 * mx_chunk(l >> 4) + j;
        const int k = (l & 15) * sh.M + sh.tpad - t;
        atab[e] = (k >= 0 && k < sh.T) ? digits[p * sh.T + k] : (signed char)0;
    }
    for (int k = tid; k < sh.T; k += MX_THREADS) gd_lds[k] = gd[k];

    // the walk: tile (c, t) = channel c, tile t of the channel; a workgroup advances by gridDim.x tiles, channel by channel
    // (no division in the loop: gridDim.x = cdiv tiles_per_ch + crem is split once)
    const int cdiv = (int)(gridDim.x / (unsigned)sh.tiles_per_ch), crem = (int)(gridDim.x % (unsigned)sh.tiles_per_ch);
    // (ioff / ooff: element offsets of the tile's first input sample, without the -tpad, and of its first output, kept by
    //  increments as well: the 64-bit products c * pitch + t * tile per tile cost ~60 scalar instructions)
    const long tile_in = (long)TILE_OUT * sh.M;
    const long in_step = (long)cdiv * in_pitch + (long)crem * tile_in, in_wrap = in_pitch - (long)sh.tiles_per_ch * tile_in;
    const long out_step = (long)cdiv * out_pitch + (long)crem * TILE_OUT, out_wrap = out_pitch - (long)sh.tiles_per_ch * TILE_OUT;
    auto advance = [&](int &c, int &t, long &ioff, long &ooff) {
        c += cdiv;
        t += crem;
        ioff += in_step;
        ooff += out_step;
        if (t >= sh.tiles_per_ch) { t -= sh.tiles_per_ch; c++; ioff += in_wrap; ooff += out_wrap; }
    };
    auto first_of = [&](int t) { return (long)t * tile_in - sh.tpad; };
    // streamed form: all but the first and the last tile of a channel (those two are read sample by sample: history in front
    // of the frame, zeros behind it)
    const int t_stream_end = (int)((n_in - total + sh.tpad) / tile_in);          // largest t with first + total <= n_in
    auto streams = [&](int c, int t) { return c < channels && aligned_in && t >= 1 && t <= t_stream_end; };
    // per-lane byte offsets of a request's 16-byte groups (the threads past the end re-read the last group)
    int poff[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int p = (j * MX_THREADS + tid) * 8;
        poff[j] = 2 * (p < last8 ? p : last8);
    }

    // front half of a tile: its samples into the LDS planes
    MX_T0();
    [[maybe_unused]] int mx_tiles = 0;
    auto stage = [&](int c, int t, i16x8 (&v)[NV], bool streamed, int young) {
        __syncthreads();                                  // the previous tile's readers of the planes are done
        MX_MARK(0);                                       // first barrier
        if (streamed) {
            // the stores of the previous tile are the only operations younger than the request
            if (young >= NACC) mx_wait_vm<NACC>();
            else mx_wait_vm<0>();
            mx_pin(v);
#pragma unroll
            for (int j = 0; j < NV; j++) {
                // 8 samples = 4 dwords; v_perm_b32 gathers the low / high bytes of four samples at a time, and
                // (x & 255) - 128 as a signed byte is the low byte with its top bit flipped
                const u32x4 d = __builtin_bit_cast(u32x4, v[j]);
                u32x2 lo, hi;
                lo[0] = __builtin_amdgcn_perm(d[1], d[0], 0x06040200u) ^ 0x80808080u;
                lo[1] = __builtin_amdgcn_perm(d[3], d[2], 0x06040200u) ^ 0x80808080u;
                hi[0] = __builtin_amdgcn_perm(d[1], d[0], 0x07050301u);
                hi[1] = __builtin_amdgcn_perm(d[3], d[2], 0x07050301u);
                *reinterpret_cast<u32x2 *>(&xs_lo[poff[j] >> 1]) = lo;
                *reinterpret_cast<u32x2 *>(&xs_hi[poff[j] >> 1]) = hi;
            }
        } else {
            const long first = first_of(t);
            const short *row = in + (size_t)c * in_pitch;
            const short *hrow = hist ? hist + (size_t)c * (sh.T - 1) : nullptr;
            for (int p = tid; p < total; p += MX_THREADS) {
                const long idx = first + p;
                int x = 0;
                if (idx >= 0) {
                    if (idx < n_in) x = row[idx];
                } else if (hrow && idx >= -(long)(sh.T - 1)) {
                    x = hrow[sh.T - 1 + idx];
                }
                xs_lo[p] = (signed char)((x & 255) - 128);
                xs_hi[p] = (signed char)(x >> 8);
            }
        }
        // (Digital silence needs no flag: every output of a silent tile is the value 0 exactly, which scr_near_zero settles
        //  without a second look.  Rounds 2-3 kept a "some sample is non-zero" word per tile in LDS and skipped the arithmetic.)
        MX_MARK(1);                                       // request awaited, planes written
        __syncthreads();
        MX_MARK(2);                                       // second barrier
    };

    // start values of the accumulators (screen_i8.hpp: the bias and the 2^31 offset ride along in the first products)
    i32x4 start[3];
    {
        int s0, s2, s4;
        scr_start(pr.bq_lo, pr.bq_hi, s0, s2, s4);
        start[0] = (i32x4){s0, s0, s0, s0};
        start[1] = (i32x4){s2, s2, s2, s2};
        start[2] = (i32x4){s4, s4, s4, s4};
    }

    // lane constants of the back half: byte offset of the lane's B operand inside a plane and of its 8 output bytes inside
    // the tile, per accumulator block
    int b_lane[NACC];
    unsigned o_lane[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        b_lane[a] = ((wave * NACC + a) * 16 + seg) * 16 * sh.M + 16 * mx_chunk(kq);
        o_lane[a] = 2u * (unsigned)(((wave * NACC + a) * 16 + seg) * 16 + 4 * kq);
    }

    // back half: products, decisions, stores; returns the number of store instructions issued per lane (for the wait count)
    auto finish = [&](int t, long ooff) {
        const long o0 = (long)t * TILE_OUT;
        short *otile = out + ooff;                                   // wave-uniform: the tile's first output
        const bool whole = aligned_out && o0 + TILE_OUT <= n_out;
        auto store4 = [&](int a, const int (&r4)[4]) {
            const int oo = ((wave * NACC + a) * 16 + seg) * 16 + 4 * kq;        // first of this lane's 4 outputs of block a
            const i16x4 y = scr_clamp4(r4);
            if (whole) {
                // (SGPR base + the lane's constant offset: no address arithmetic per store)
                asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(o_lane[a]), "v"(y), "s"(otile) : "memory");
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (o0 + oo + j < n_out) otile[oo + j] = y[j];
            }
        };
        i32x4 acc[NACC][MX_ACCS];
        {
            const signed char *bp[NACC], *bph[NACC];
#pragma unroll
            for (int a = 0; a < NACC; a++) {
                bp[a] = xs_lo + b_lane[a];
                bph[a] = xs_hi + b_lane[a];
            }
            const signed char *ap = atab + lane * 16;
            // one step = 64 window samples: tap planes 0..4 against the high sample plane (weights 1..5), planes 1..4 against
            // the low one (weights 1..4).  The first step starts every accumulator from the constant 0 (no zeroing pass).
            auto step = [&](int s, auto first_step) {
                i32x4 ad[MX_PLANES], bd[NACC][2];
#pragma unroll
                for (int p = 0; p < MX_PLANES; p++) ad[p] = *reinterpret_cast<const i32x4 *>(ap + (s * MX_PLANES + p) * 1024);
#pragma unroll
                for (int a = 0; a < NACC; a++) {
                    bd[a][0] = *reinterpret_cast<const i32x4 *>(bp[a] + s * 64);
                    bd[a][1] = *reinterpret_cast<const i32x4 *>(bph[a] + s * 64);
                }
#pragma unroll
                for (int p = 0; p < MX_PLANES; p++)
#pragma unroll
                    for (int d = 1; d >= 0; d--)
#pragma unroll
                        for (int a = 0; a < NACC; a++) {
                            if (d == 0 && p == 0) continue;          // low sample digit x lowest tap digit: not formed (inside eps)
                            // accumulator p + d - 1; in this order (p, d = 1) is always its first visitor
                            // (and starts it: from scr_start's values for accumulators 0, 2, 4, from the constant 0 otherwise)
                            const bool fresh = decltype(first_step)::value && d == 1;
                            const i32x4 c0 = !fresh ? acc[a][p + d - 1] : ((p & 1) ? (i32x4){0, 0, 0, 0} : start[p >> 1]);
                            acc[a][p + d - 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ad[p], bd[a][d], c0, 0, 0, 0);
                        }
            };
            step(0, std::true_type{});
            if constexpr (KS > 0) {
                if constexpr (KS > 1) step(1, std::false_type{});
            } else {
                for (int s = 1; s < ksteps; s++) step(s, std::false_type{});
            }
        }

        // every output decided without a branch; a lane notes the slots it could not decide in a bit mask, and the rare lanes
        // that have any (about 2 eps of all outputs: most tiles have none) work them off one by one afterwards
#ifdef MX_TRACE
        asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[NACC - 1][4][0]));
#endif
        MX_MARK(4);                                                 // operand reads, products
        int res[NACC][4];
        bool unsure[NACC][4];
        unsigned long long open = 0;                                // lanes with an undecided slot (a scalar mask)
#pragma unroll
        for (int a = 0; a < NACC; a++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                res[a][j] = scr_decide<NEG>(acc[a][0][j], acc[a][1][j], acc[a][2][j], acc[a][3][j], acc[a][4][j], pr.rs, pr.e32,
                                            unsure[a][j]);
                open |= __ballot(unsure[a][j]);
            }
        if (open != 0) {
            unsigned mine = 0;
#pragma unroll
            for (int a = 0; a < NACC; a++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    // (not a value within eps of zero: that truncates to 0 from either side -- screen_i8.hpp)
                    const bool look = unsure[a][j] && !scr_near_zero<NEG>(acc[a][0][j], acc[a][1][j], acc[a][2][j], acc[a][3][j],
                                                                          acc[a][4][j], pr.rs, pr.e32);
                    mine = mine + mine + (look ? 1u : 0u);                         // slot 4 a + j at bit 4 NACC - 1 - (4 a + j)
                }
#pragma unroll 1
            while (mine != 0) {
                // an integer within eps of the value: the reference's own arithmetic decides, in the lane that found it
                const int slot = 4 * NACC - 1 - __builtin_ctz(mine);
                mine &= mine - 1;
                const int oo = ((wave * NACC + (slot >> 2)) * 16 + seg) * 16 + 4 * kq + (slot & 3);
                int cur = 0;
#pragma unroll
                for (int a = 0; a < NACC; a++)
#pragma unroll
                    for (int j = 0; j < 4; j++) cur = slot == 4 * a + j ? res[a][j] : cur;
                // (a value far outside the clamp range needs no second look: its int16 is the rail either way)
                if (o0 + oo < n_out && scr_in_reach(cur)) {
                    const int r = mx_exact(xs_hi, xs_lo, oo * sh.M + sh.tpad, gd_lds, sh.T, pr.gain);
#pragma unroll
                    for (int a = 0; a < NACC; a++)
#pragma unroll
                        for (int j = 0; j < 4; j++) res[a][j] = slot == 4 * a + j ? r : res[a][j];
                }
            }
        }
#ifdef MX_TRACE
        asm volatile("s_nop 0" ::"v"(res[0][0]), "v"(res[NACC - 1][3]));
#endif
        MX_MARK(5);                                                 // decisions (and second looks)
#pragma unroll
        for (int a = 0; a < NACC; a++) store4(a, res[a]);
        MX_MARK(6);                                                 // stores
        return whole ? NACC : 0;
    };

    // One loop, one request site.  Iteration k stages and finishes tile k (none in the first iteration) and requests tile
    // k + 1 between the two halves, right behind the staging barrier: the request has a whole tile's arithmetic to arrive.
    int c = 0, t = 0;                                   // the tile in work (none yet)
    long ooff = 0;
    int cn = (int)(blockIdx.x / (unsigned)sh.tiles_per_ch), tn = (int)(blockIdx.x % (unsigned)sh.tiles_per_ch);
    long ioff_n = (long)cn * in_pitch + (long)tn * tile_in, ooff_n = (long)cn * out_pitch + (long)tn * TILE_OUT;
    bool in_work = false, streamed = false;
    int young = 0;
    i16x8 v[NV];
    while (in_work || cn < channels) {
        if (in_work) stage(c, t, v, streamed, young);
        const bool next_streams = streams(cn, tn);
        // (a tile that does not stream requests the first row instead: in bounds -- the launcher requires n_in >= total --
        //  and never looked at)
        const short *src = next_streams ? in + (ioff_n - sh.tpad) : in;
#pragma unroll
        for (int j = 0; j < NV; j++) v[j] = mx_load_nt(src, poff[j]);
        MX_MARK(3);                                         // next tile requested
        young = 0;
        if (in_work) { young = finish(t, ooff); mx_tiles++; }
        c = cn;
        t = tn;
        ooff = ooff_n;
        in_work = c < channels;
        streamed = next_streams;
        advance(cn, tn, ioff_n, ooff_n);
    }
#ifdef MX_TRACE
    mx_acc[7] = (unsigned long long)mx_tiles;
#endif
    MX_DUMP();
}

bool mx_make_shape(int T, int M, int nacc, long n_out, mx_shape *sh, size_t *lds_bytes)
{
    sh->T = T;
    sh->M = M;
    sh->tpad = (T - 1 + 7) & ~7;
    sh->ksteps = (sh->tpad + 15 * M + 1 + 63) / 64;
    const int tile_out = MX_WAVES * nacc * 256;
    sh->total = ((tile_out - 16) * M + 64 * sh->ksteps + 7) & ~7;
    sh->plane = (sh->total + 16 + 15) & ~15;
    sh->tiles_per_ch = (int)((n_out + tile_out - 1) / tile_out);
    *lds_bytes = (size_t)MX_PLANES * sh->ksteps * 1024 + 2 * (size_t)sh->plane + sizeof(double) * (size_t)T;
    // pairs of plane sums are combined in 32 bits: (2 T 2^14)(256 + 1) < 2^31 needs T <= 200; the staging needs
    // total <= NV_MAX x 256 x 8 samples
    return *lds_bytes <= 160 * 1024 && sh->total <= MX_NV_MAX * MX_THREADS * 8 && T <= 200;
}

int mx_pick_nacc(int T, int M)
{
    mx_shape sh;
    size_t bytes;
    if (mx_make_shape(T, M, 2, 1, &sh, &bytes) && bytes <= 78 * 1024) return 2;
    if (mx_make_shape(T, M, 1, 1, &sh, &bytes)) return 1;
    return 0;
}

template <int NACC, int NV, bool NEG, int KS>
int mx_launch(const short *in, short *out, const short *hist, const signed char *digits, const double *gd, int channels,
              long n_in, long n_out, long in_pitch, long out_pitch, int T, int M, const mx_params &pr, void *stream)
{
    mx_shape sh;
    size_t lds_bytes;
    mx_make_shape(T, M, NACC, n_out, &sh, &lds_bytes);
    if (lds_bytes > 64 * 1024)
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir_mfma_i8x<NACC, NV, NEG, KS>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const long ntiles = (long)sh.tiles_per_ch * channels;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
        (void)hipGetLastError();
        cus = 256;
    }
    // a persistent grid: exactly the workgroups the chip holds at once (a workgroup that has to wait for a slot would
    // run its share of the tiles after everyone else: 768 workgroups on 512 slots measured 48 ms instead of 36)
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(k_fir_mfma_i8x<NACC, NV, NEG, KS>),
                                                     MX_THREADS, lds_bytes) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 2;
    }
    if (const int v = llzs_tune(LLZS_TUNE_MFMA_WG_PER_CU); v >= 1 && v <= 8) per_cu = v;
    long grid = (long)cus * per_cu;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((k_fir_mfma_i8x<NACC, NV, NEG, KS>), dim3((unsigned)grid), dim3(MX_THREADS), lds_bytes, as_stream(stream),
                       in, out, hist, digits, gd, n_in, n_out, in_pitch, out_pitch, sh, pr, channels);
    LLZ_LAUNCH_CHECK("k_fir_mfma_i8x");
    return LLZ_OK;
}

} // namespace

extern "C" int llzs_fir_mfma_i16x_fits(int T, int M)
{
    return T >= 1 && M >= 1 && mx_pick_nacc(T, M) > 0;
}

extern "C" int llzs_fir_mfma_i16x(const short *in, short *out, const short *hist, const signed char *digits,
                                  const double *gd, int channels, long n_in, long n_out, long in_pitch, long out_pitch,
                                  int T, int M, int shift, long long bias, double gain, double eps, void *stream)
{
    if (!in || !out || !digits || !gd || channels <= 0 || n_in <= 0 || n_out <= 0 || T < 1 || M < 1 || in_pitch < n_in ||
        out_pitch < n_out || (n_out - 1) * M >= n_in || shift < 32 || shift > 46 || !(eps > 0.0) || !(eps < 0.0625)) {
        llzs_set_error("fir_mfma_i16x: bad arguments (channels=%d n_in=%ld n_out=%ld T=%d M=%d shift=%d eps=%g)", channels,
                       n_in, n_out, T, M, shift, eps);
        return LLZ_ERR_ARG;
    }
    const int nb = mx_pick_nacc(T, M);
    if (!nb) {
        llzs_set_error("fir_mfma_i16x: %d taps at decimation %d do not fit the LDS image", T, M);
        return LLZ_ERR_RANGE;
    }
    mx_params pr;
    const long long bq = bias >= 0 ? bias / 256 : -((-bias + 255) / 256);        // floor(bias / 256)
    pr.bq_lo = (int)(unsigned)((unsigned long long)bq & 0xffffffffull);
    pr.bq_hi = (int)(bq >> 32);
    pr.rs = shift - 40;
    pr.e32 = (unsigned)ceil(ldexp(eps, 32)) + 2u;
    pr.gain = gain;
    mx_shape sh;
    size_t bytes;
    mx_make_shape(T, M, nb, n_out, &sh, &bytes);
    // the kernel's sample requests are unconditional: a tile that cannot stream reads the head of the first row instead, so
    // that head must exist and be aligned (frames shorter than one tile, odd pitches: the caller takes the all-double kernel)
    if (n_in < sh.total || (in_pitch & 7) != 0 || (reinterpret_cast<uintptr_t>(in) & 15) != 0) {
        llzs_set_error("fir_mfma_i16x: frame of %ld samples (pitch %ld) is too short or not 16-byte aligned for the tile of %d",
                       n_in, in_pitch, sh.total);
        return LLZ_ERR_RANGE;
    }
    const bool small = sh.total <= 4 * MX_THREADS * 8;
#define MX_GO3(A, V, N, K) return mx_launch<A, V, N, K>(in, out, hist, digits, gd, channels, n_in, n_out, in_pitch, out_pitch, T, M, pr, stream)
#define MX_GO2(A, V, N)                                                                                                        \
    do {                                                                                                                       \
        if (sh.ksteps == 1) MX_GO3(A, V, N, 1);                                                                                \
        if (sh.ksteps == 2) MX_GO3(A, V, N, 2);                                                                                \
        MX_GO3(A, V, N, 0);                                                                                                    \
    } while (0)
#define MX_GO(A, V)                                                                                                            \
    do {                                                                                                                       \
        if (pr.rs < 0) MX_GO2(A, V, true);                                                                                     \
        MX_GO2(A, V, false);                                                                                                   \
    } while (0)
    if (nb == 2) { if (small) MX_GO(2, 4); else MX_GO(2, 8); }
    if (small) MX_GO(1, 4); else MX_GO(1, 8);
#undef MX_GO
#undef MX_GO2
#undef MX_GO3
}

#ifdef MX_TRACE
extern "C" int llzs_mx_trace_read(unsigned long long *dst, int count)
{
    LLZ_HIP_CHECK(hipDeviceSynchronize());
    LLZ_HIP_CHECK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(mx_trace_buf), sizeof(unsigned long long) * (size_t)count));
    return LLZ_OK;
}
#endif
