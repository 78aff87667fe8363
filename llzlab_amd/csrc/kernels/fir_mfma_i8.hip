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
//      the ten digit-plane products in int32 -- exactly: |sum| <= 2 Q 2^14 -- into six accumulators by weight w = p + a, and
//            S = sum_w 256^w acc_w + 128 sum_k G[k]        (int64, exact)       = sum_k x[iM - k] G[k]
//      (the gain is folded into G: G[k] = round(gain g[k] 2^s)), so v = S 2^-s differs from the reference's y only by the
//      tap quantisation and the reference's own rounding:
//            |v - y| <= eps := 2 (32768 sum_k |gain g[k] - G[k] 2^-s|  +  |gain| (Q + 2) 2^-52 32768 sum_k |g[k]|) + 2^-30
//      (the host evaluates this for the handle's taps: ~4e-6 at Q = 134; DESIGN.md has the derivation).
//   2. DECIDE, in integers.  With E = ceil(eps 2^s): if the fractional field of S + E (s bits) exceeds 2 E, no integer
//      lies within eps of v, so v and y truncate (and clamp) to the same int16: S >> s, plus one if negative.
//   3. RECOMPUTE the others -- about 2 eps of all outputs, ~1e-5 -- in the reference's exact order, in double, by the lane
//      that found them, from the tile's samples in LDS (rounded multiply, rounded add, ascending k: the loop above).
//      All-zero tiles (digital silence: every output would sit ON the integer 0) are written as zeros without arithmetic.
//
// The result is the reference's int16 for every sample by construction; the tests compare it bit for bit with the oracle on
// random, clipping, silent and DC inputs.  Mapping and data movement are fir_mfma.hip's (banded-Toeplitz product: a wave
// owns 16 consecutive 16-output segments of a channel, D[m][n] = sum_t A[m][t] B[t][n] with A[m][t] = G[mM + tpad - t],
// B[t][n] = sample t of segment n's window; persistent workgroups walk tiles, the next tile's samples prefetched into
// registers), with 64 window samples per MFMA step and one-byte planes in LDS.
#include "common.hpp"
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
constexpr int MX_NV_MAX = 8;                // prefetch registers: 8 x 16 B (8 samples each) per thread

struct mx_shape {
    int T, M;
    int tpad;        // multiple of 8 >= T-1 (int16 rows stay 16-byte aligned at tile starts)
    int ksteps;      // MFMA steps of 64 window samples
    int total;       // samples staged per tile (multiple of 8)
    int plane;       // bytes per LDS sample plane (multiple of 16)
    int tiles_per_ch;
};

struct mx_params {
    long long bias;          // 128 * sum_k G[k]: the samples' +128 offset
    unsigned long long E;    // ceil(eps 2^shift): the screen's uncertainty in units of 2^-shift
    unsigned thr;            // (2 E >> (shift - 32)) + 1
    int shift;               // 32 .. 46: S 2^-shift is the screen's value of the output (the taps carry the gain)
    double gain;             // only the recompute path multiplies by it, as the reference does
};

// the reference's loop for one output, from the tile's LDS image: sample p of the window is 256 hi[p] + lo[p] + 128
__device__ __noinline__ short mx_exact(const signed char *hi, const signed char *lo, int p0, const double *__restrict__ gd,
                                        int T, double gain)
{
#pragma clang fp contract(off)
    double y = 0.0;
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

template <int NACC, int NV>
__global__ void __launch_bounds__(MX_THREADS)
k_fir_mfma_i8x(const short *__restrict__ in, short *__restrict__ out, const short *__restrict__ hist,
               const signed char *__restrict__ digits /* [MX_PLANES][T] */, const double *__restrict__ gd, long n_in,
               long n_out, long in_pitch, long out_pitch, mx_shape sh, mx_params pr, long ntiles)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int TILE_OUT = MX_WAVES * NACC * 256;
    const int aplane = sh.ksteps * 1024;                 // A table: [planes][ksteps][64 lanes][16 bytes]
    signed char *atab = reinterpret_cast<signed char *>(lds);
    signed char *xs_lo = atab + MX_PLANES * aplane;      // sample planes: low digit (x & 255) - 128, then x >> 8
    signed char *xs_hi = xs_lo + sh.plane;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int total = sh.total, last8 = sh.total - 8;
    const bool aligned_in = (in_pitch & 7) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const bool aligned_out = (out_pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0;

    for (int e = tid; e < MX_PLANES * aplane; e += MX_THREADS) {
        const int p = e / aplane, r = e - p * aplane;
        const int s = r >> 10, l = (r >> 4) & 63, j = r & 15;
        const int t = 64 * s + 16 * (l >> 4) + j;
        const int k = (l & 15) * sh.M + sh.tpad - t;
        atab[e] = (k >= 0 && k < sh.T) ? digits[p * sh.T + k] : (signed char)0;
    }

    auto tile_first = [&](long q, int &c, long &o0) {
        c = (int)(q / sh.tiles_per_ch);
        o0 = (q - (long)c * sh.tiles_per_ch) * TILE_OUT;
        return o0 * sh.M - sh.tpad;
    };
    auto is_interior = [&](long first) { return aligned_in && first >= 0 && first + total <= n_in; };

    auto prefetch = [&](i16x8 (&v)[NV], long q) {
        int c; long o0;
        const long first = tile_first(q, c, o0);
        if (!is_interior(first)) return false;
        const short *src = in + (size_t)c * in_pitch + first;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            if (j * MX_THREADS * 8 < total) {
                int p = (j * MX_THREADS + tid) * 8;
                p = p < last8 ? p : last8;
                v[j] = __builtin_nontemporal_load(reinterpret_cast<const i16x8 *>(src + p));
            }
        }
        return true;
    };

    auto tile = [&](long q, i16x8 (&v)[NV], bool &have) {
        int c; long o0;
        const long first = tile_first(q, c, o0);
        __syncthreads();
        int nonzero = 0;
        if (have) {
#pragma unroll
            for (int j = 0; j < NV; j++) {
                if (j * MX_THREADS * 8 < total) {
                    const int p = (j * MX_THREADS + tid) * 8;
                    // 8 samples = 4 dwords; v_perm_b32 gathers the low / high bytes of four samples at a time, and
                    // (x & 255) - 128 as a signed byte is the low byte with its top bit flipped
                    const u32x4 d = __builtin_bit_cast(u32x4, v[j]);
                    nonzero |= (int)(d[0] | d[1] | d[2] | d[3]);
                    u32x2 lo, hi;
                    lo[0] = __builtin_amdgcn_perm(d[1], d[0], 0x06040200u) ^ 0x80808080u;
                    lo[1] = __builtin_amdgcn_perm(d[3], d[2], 0x06040200u) ^ 0x80808080u;
                    hi[0] = __builtin_amdgcn_perm(d[1], d[0], 0x07050301u);
                    hi[1] = __builtin_amdgcn_perm(d[3], d[2], 0x07050301u);
                    if (p < total) {
                        *reinterpret_cast<u32x2 *>(&xs_lo[p]) = lo;
                        *reinterpret_cast<u32x2 *>(&xs_hi[p]) = hi;
                    }
                }
            }
        } else {
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
                nonzero |= x;
                xs_lo[p] = (signed char)((x & 255) - 128);
                xs_hi[p] = (signed char)(x >> 8);
            }
        }
        const int any = __syncthreads_or(nonzero);
        have = q + gridDim.x < ntiles && prefetch(v, q + gridDim.x);

        short *orow = out + (size_t)c * out_pitch;
        const bool whole = aligned_out && o0 + TILE_OUT <= n_out;
        i32x4 acc[NACC][MX_PLANES + 1];
        if (any) {
            const signed char *bp[NACC];
#pragma unroll
            for (int a = 0; a < NACC; a++) bp[a] = xs_lo + ((wave * NACC + a) * 16 + n) * 16 * sh.M + 16 * kq;
            const signed char *ap = atab + lane * 16;
            // one step = 64 window samples: 5 tap planes x 2 sample planes.  The first step starts every accumulator from
            // the constant 0 (no zeroing pass); the other resident waves cover the LDS latency of a step's operands
            auto step = [&](int s, auto first) {
                i32x4 ad[MX_PLANES], bd[NACC][2];
#pragma unroll
                for (int p = 0; p < MX_PLANES; p++) ad[p] = *reinterpret_cast<const i32x4 *>(ap + p * aplane + s * 1024);
#pragma unroll
                for (int a = 0; a < NACC; a++) {
                    bd[a][0] = *reinterpret_cast<const i32x4 *>(bp[a] + s * 64);
                    bd[a][1] = *reinterpret_cast<const i32x4 *>(bp[a] + sh.plane + s * 64);
                }
#pragma unroll
                for (int p = 0; p < MX_PLANES; p++)
#pragma unroll
                    for (int d = 0; d < 2; d++)
#pragma unroll
                        for (int a = 0; a < NACC; a++) {
                            // weight p + d is first written by (p = 0, d = 0), (p, d = 1) for p < 5 ... in this loop order:
                            // (p, d) is the first visitor of p + d exactly when d == 1 or p == 0
                            const bool fresh = decltype(first)::value && (d == 1 || p == 0);
                            const i32x4 c = fresh ? (i32x4){0, 0, 0, 0} : acc[a][p + d];
                            acc[a][p + d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ad[p], bd[a][d], c, 0, 0, 0);
                        }
            };
            step(0, std::true_type{});
            for (int s = 1; s < sh.ksteps; s++) step(s, std::false_type{});
        }

#pragma unroll
        for (int a = 0; a < NACC; a++) {
            const int oo = ((wave * NACC + a) * 16 + n) * 16 + 4 * kq;        // first of this lane's 4 outputs in the tile
            const long o = o0 + oo;
            i16x4 y;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                short r = 0;
                if (any) {
                    // S = sum_w 256^w acc_w + bias, exact: the plane sums pair up in 32 bits (|acc_w| <= 2 T 2^14, T <= 200)
                    const int p01 = acc[a][0][j] + acc[a][1][j] * 256;
                    const int p23 = acc[a][2][j] + acc[a][3][j] * 256;
                    const int p45 = acc[a][4][j] + acc[a][5][j] * 256;
                    long long S = (long long)p23 * 65536 + ((long long)p01 + pr.bias);
                    S += (long long)((unsigned long long)(unsigned)p45 << 32);
                    const int I = (int)(S >> pr.shift);                                   // floor(v)
                    const unsigned frac = (unsigned)(((unsigned long long)S + pr.E) >> (pr.shift - 32));
                    if (frac <= pr.thr && (unsigned)(I + 32770) <= 65540u) {
                        // an integer within eps of v (and v inside the clamp range): the reference's own arithmetic decides
                        r = (o + j < n_out) ? mx_exact(xs_hi, xs_lo, (oo + j) * sh.M + sh.tpad, gd, sh.T, pr.gain) : (short)0;
                    } else {
                        int t = I + (int)((unsigned long long)S >> 63);                   // toward zero: v is not an integer here
                        t = t > 32767 ? 32767 : t;
                        t = t < -32768 ? -32768 : t;
                        r = (short)t;
                    }
                }
                y[j] = r;
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
    i16x8 v[NV];
    bool have = q < ntiles && prefetch(v, q);
    for (; q < ntiles; q += gridDim.x) tile(q, v, have);
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
    *lds_bytes = (size_t)MX_PLANES * sh->ksteps * 1024 + 2 * (size_t)sh->plane;
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

template <int NACC, int NV>
int mx_launch(const short *in, short *out, const short *hist, const signed char *digits, const double *gd, int channels,
              long n_in, long n_out, long in_pitch, long out_pitch, int T, int M, const mx_params &pr, void *stream)
{
    mx_shape sh;
    size_t lds_bytes;
    mx_make_shape(T, M, NACC, n_out, &sh, &lds_bytes);
    if (lds_bytes > 64 * 1024)
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir_mfma_i8x<NACC, NV>),
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
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(k_fir_mfma_i8x<NACC, NV>),
                                                     MX_THREADS, lds_bytes) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 2;
    }
    if (const int v = llzs_tune(LLZS_TUNE_MFMA_WG_PER_CU); v >= 1 && v <= 8) per_cu = v;
    long grid = (long)cus * per_cu;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((k_fir_mfma_i8x<NACC, NV>), dim3((unsigned)grid), dim3(MX_THREADS), lds_bytes, as_stream(stream), in,
                       out, hist, digits, gd, n_in, n_out, in_pitch, out_pitch, sh, pr, ntiles);
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
        out_pitch < n_out || (n_out - 1) * M >= n_in || shift < 32 || shift > 46 || !(eps > 0.0) || !(eps < 0.125)) {
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
    pr.bias = bias;
    pr.shift = shift;
    pr.E = (unsigned long long)ceil(ldexp(eps, shift)) + 1;
    pr.thr = (unsigned)((2 * pr.E) >> (shift - 32)) + 1;
    pr.gain = gain;
    mx_shape sh;
    size_t bytes;
    mx_make_shape(T, M, nb, n_out, &sh, &bytes);
    const bool small = sh.total <= 4 * MX_THREADS * 8;
#define MX_GO(A, V) return mx_launch<A, V>(in, out, hist, digits, gd, channels, n_in, n_out, in_pitch, out_pitch, T, M, pr, stream)
    if (nb == 2) { if (small) MX_GO(2, 4); else MX_GO(2, 8); }
    if (small) MX_GO(1, 4); else MX_GO(1, 8);
#undef MX_GO
}
