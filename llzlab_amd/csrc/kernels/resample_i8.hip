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
// legal on gfx950 (it is what hipcc itself emits for an under-aligned load), and no second copy of any sample is needed (a
// first version kept one aligned column per period: 1.7 copies of every sample, byte stores when M is odd -- 3.6 ms against
// 1.2 ms at 160:147 / 147:160).  Results go to an LDS image of the output in memory order and leave in 8-byte pieces; the next
// span's samples are requested one span ahead into registers.
#include "screen_i8.hpp"
#include <math.h>

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
    int spans_per_wg;
    int rs;                 // shift - 40
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

template <int KS, bool NEG, bool RELOAD>
__global__ void __launch_bounds__(512)
k_resample_i8x(const short *__restrict__ in, short *__restrict__ out, const short *__restrict__ hist,
               const signed char *__restrict__ atab /* [nt][KS][5][64][16] */, const int *__restrict__ aoff /* [nt] */,
               const int *__restrict__ bqtab /* [16 nt][4]: bias lo, hi; e32; first | last << 8 non-zero tap | exact << 16 */, const double *__restrict__ g /* [L][Q] */, long n_in, long n_out,
               long in_pitch, long out_pitch, ri_shape sh)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    signed char *xs_lo = reinterpret_cast<signed char *>(lds);
    signed char *xs_hi = xs_lo + sh.plane;
    short *oimg = reinterpret_cast<short *>(xs_hi + sh.plane);          // [P][L] outputs in memory order
    const int tid = threadIdx.x, lane = tid & 63, threads = (int)blockDim.x, waves = threads >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int col16 = scr_col(n), ck = scr_chunk(kq);
    const int c = blockIdx.y;
    const short *row = in + (size_t)c * in_pitch;
    const short *hrow = hist ? hist + (size_t)c * (sh.Q - 1) : nullptr;
    short *orow = out + (size_t)c * out_pitch;
    const bool vec_in = (in_pitch & 7) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const bool vec_out = (out_pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0;

    // a span's samples: from absolute index M m0 - Hp on, requested in 16-byte groups starting at the 8-aligned index at or below it
    i16x8 v[RI_NG];
    auto request = [&](long m0s) {
        const long s0 = m0s * sh.M - sh.Hp;
        const long gbase = s0 >= 0 ? (s0 & ~7L) : -((-s0 + 7) & ~7L);
#pragma unroll
        for (int k = 0; k < RI_NG; k++) {
            const int gi = k * threads + tid;
            i16x8 w = {0, 0, 0, 0, 0, 0, 0, 0};
            if (gi < sh.ngroups) {
                const long a = gbase + 8L * gi;
                if (vec_in && a >= 0 && a + 8 <= n_in) {
                    w = __builtin_nontemporal_load(reinterpret_cast<const i16x8 *>(row + a));
                } else {
                    // a frame edge (history in front, zeros behind) or an unaligned frame: sample by sample, packed into dwords
                    unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0;
#pragma unroll 1
                    for (int e = 0; e < 8; e++) {
                        const long idx = a + e;
                        int x = 0;
                        if (idx >= 0) {
                            if (idx < n_in) x = row[idx];
                        } else if (hrow && idx >= -(long)(sh.Q - 1)) {
                            x = hrow[sh.Q - 1 + idx];
                        }
                        const unsigned bits = ((unsigned)x & 0xffffu) << (16 * (e & 1));
                        d0 |= (e >> 1) == 0 ? bits : 0u;
                        d1 |= (e >> 1) == 1 ? bits : 0u;
                        d2 |= (e >> 1) == 2 ? bits : 0u;
                        d3 |= (e >> 1) == 3 ? bits : 0u;
                    }
                    w = __builtin_bit_cast(i16x8, (u32x4){d0, d1, d2, d3});
                }
            }
            v[k] = w;
        }
    };

    scr_i32x4 ad[KS][5];
    auto load_a = [&](int t) {
        const signed char *ap = atab + ((size_t)t * KS * 5) * 1024 + lane * 16;
#pragma unroll
        for (int s = 0; s < KS; s++)
#pragma unroll
            for (int p = 0; p < 5; p++) ad[s][p] = *reinterpret_cast<const scr_i32x4 *>(ap + (s * 5 + p) * 1024);
    };
    if (!RELOAD && wave < sh.nt) load_a(wave);

    const int P = sh.P;
    const long span0 = (long)blockIdx.x * sh.spans_per_wg;
    const long span1 = min(span0 + sh.spans_per_wg, sh.spans);
    if (span0 < span1) request(span0 * P);
    for (long sp = span0; sp < span1; sp++) {
        const long m0 = sp * P;
        const long s0 = m0 * sh.M - sh.Hp;
        // the image starts at the 8-aligned sample index at or below s0: window position pos of period q sits `lead + M q + pos`
        // bytes into a plane
        const int lead = (int)(s0 - (s0 >= 0 ? (s0 & ~7L) : -((-s0 + 7) & ~7L)));
        // ---- the requested groups into the planes (the previous span's readers passed the barrier at its end) ----
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
                *reinterpret_cast<u32x2 *>(&xs_lo[8 * gi]) = lo;
                *reinterpret_cast<u32x2 *>(&xs_hi[8 * gi]) = hi;
            }
        }
        if (sp + 1 < span1) request(m0 + P);
        __syncthreads();

        // ---- products and decisions: this wave's phase tile(s) x the span's period tiles ----
        const long periods_left = (n_out - m0 * sh.L + sh.L - 1) / sh.L;          // periods of this span that exist
        for (int t = wave; t < sh.nt; t += waves) {
            if (RELOAD) load_a(t);
            const int a_t = aoff[t];
            const int f0 = 16 * t + 4 * kq;                                       // the lane's first phase (row 4 kq) of the tile
            int bql[4], bqh[4];
            unsigned e32[4], never = 0;                                            // never: slots of phases that cannot be unsure
            int krange[4];                                                         // first | last << 8 non-zero tap of the phase
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const scr_i32x4 row4 = *reinterpret_cast<const scr_i32x4 *>(bqtab + 4 * (f0 + j));
                bql[j] = row4[0];
                bqh[j] = row4[1];
                e32[j] = (unsigned)row4[2];
                krange[j] = row4[3] & 0xffff;
                never |= (row4[3] >> 16) ? 1u << (3 - j) : 0u;
            }
#pragma unroll 1
            for (int p = 0; p < sh.pt; p++) {
                const int col = 16 * p + col16;
                const signed char *bp = xs_lo + lead + col * sh.M + a_t + 16 * ck;
                scr_i32x4 acc[5];
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    const scr_i32x4 b_lo = reinterpret_cast<const ri_b128 *>(bp + 64 * s)->v;
                    const scr_i32x4 b_hi = reinterpret_cast<const ri_b128 *>(bp + sh.plane + 64 * s)->v;
                    if (s == 0) scr_step<true>(acc, ad[s], b_lo, b_hi);
                    else scr_step<false>(acc, ad[s], b_lo, b_hi);
                }
                int res[4];
                unsigned mine = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    bool unsure;
                    res[j] = scr_decide<NEG>(acc[0][j], acc[1][j], acc[2][j], acc[3][j], acc[4][j], bql[j], bqh[j], sh.rs, e32[j],
                                             unsure, (never >> (3 - j)) & 1);
                    mine = mine + mine + (unsure ? 1u : 0u);                      // slot j at bit 3 - j
                }
                mine &= ~never;
                if (__ballot(mine != 0) != 0) {
#pragma unroll 1
                    while (mine != 0) {
                        // an integer within eps of the value: the reference's own arithmetic decides, in the lane that found it
                        const int j = 3 - __builtin_ctz(mine);
                        mine &= mine - 1;
                        const int f = f0 + j;
                        int cur = 0, kr = 0;
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            cur = j == u ? res[u] : cur;
                            kr = j == u ? krange[u] : kr;
                        }
                        // (a value far outside the clamp range needs no second look, nor does a row or period that does not exist)
                        if (f < sh.L && col < periods_left && scr_in_reach(cur)) {
                            const int cf = (int)(((long)f * sh.M) / sh.L);
                            const int r = ri_exact(xs_hi, xs_lo, lead + col * sh.M + cf + sh.Hp, g + (size_t)f * sh.Q, kr & 255, kr >> 8,
                                                   sh.gain);
#pragma unroll
                            for (int u = 0; u < 4; u++) res[u] = j == u ? r : res[u];
                        }
                    }
                }
                short *op = oimg + col * sh.L + f0;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (f0 + j < sh.L) op[j] = scr_clamp(res[j]);
            }
        }
        __syncthreads();

        // ---- the output image leaves in memory order ----
        short *ospan = orow + m0 * sh.L;
        const long left = n_out - m0 * sh.L;
        const int total = (int)(left < (long)P * sh.L ? left : (long)P * sh.L);
        if (vec_out) {                                                           // (m0 L is a multiple of 16: 8-byte pieces stay aligned)
            const int quads = total >> 2;
            for (int e = tid; e < quads; e += threads)
                *reinterpret_cast<i16x4 *>(ospan + 4 * e) = *reinterpret_cast<const i16x4 *>(oimg + 4 * e);
            for (int e = 4 * quads + tid; e < total; e += threads) ospan[e] = oimg[e];
        } else {
            for (int e = tid; e < total; e += threads) ospan[e] = oimg[e];
        }
        // (the next span's image is written by threads that have passed the barrier above: nobody reads the planes any more;
        //  the output image is next written behind the next span's staging barrier)
    }
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

// geometry of the launch
static bool ri_make_shape(int L, int M, int Q, long n_out, int channels, ri_shape *sh, int *waves, size_t *lds)
{
    sh->L = L; sh->M = M; sh->Q = Q;
    sh->nt = (L + 15) / 16;
    sh->Hp = (Q - 1 + 7) & ~7;
    const int ks = ri_ksteps(L, M, Q);
    if (ks > 4) return false;
    // waves: the phase tiles dealt evenly over at most 8 waves (147 phases: 10 tiles -> 5 waves x 2 tiles)
    const int rounds = (sh->nt + 7) / 8;
    int w = (sh->nt + rounds - 1) / rounds;
    if (w < 2) w = 2;
    *waves = w;
    bool ok = false;
    for (int pt = 4; pt >= 1 && !ok; pt--) {
        sh->pt = pt; sh->P = 16 * pt;
        // the last period's last band ends at most M - 1 + Hp + 64 ks positions into its window; + 7 of alignment slack in front
        const long bytes = 7 + (long)M * (sh->P - 1) + (M - 1) + sh->Hp + 64 * ks + 16;
        sh->ngroups = (int)((bytes + 7) / 8);
        sh->plane = (8 * sh->ngroups + 15) & ~15;
        *lds = 2 * (size_t)sh->plane + sizeof(short) * (size_t)sh->P * L;
        while (*waves < 8 && sh->ngroups > RI_NG * 64 * *waves) (*waves)++;
        ok = sh->ngroups <= RI_NG * 64 * *waves && (*lds <= 40 * 1024 || (pt == 1 && *lds <= 160 * 1024));
    }
    if (!ok) return false;
    const long periods = (n_out + L - 1) / L;
    sh->spans = (periods + sh->P - 1) / sh->P;
    // consecutive spans per workgroup: the walk length that minimises rounds x (length + 1) over ~3 resident workgroups per CU
    long spw = sh->spans < 4 ? sh->spans : 4;
    double best = 1e300;
    for (long cnt = spw; cnt <= sh->spans; cnt++) {
        const long wgs = ((sh->spans + cnt - 1) / cnt) * (long)channels;
        const double cost = (double)((wgs + 767) / 768) * (double)(cnt + 1);
        if (cost < best * 0.999) { best = cost; spw = cnt; }
    }
    sh->spans_per_wg = (int)spw;
    return true;
}

extern "C" int llzs_resample_i16x_fits(int L, int M, int Q)
{
    if (L < 2 || M < 1 || Q < 1 || Q > 200) return 0;
    ri_shape sh;
    int waves;
    size_t lds;
    return ri_make_shape(L, M, Q, L, 1, &sh, &waves, &lds) ? 1 : 0;
}

// atab: [ceil(L/16)][ksteps][5][64][16] tap digits in operand order; aoff: [ceil(L/16)] band starts; bqtab: [16 ceil(L/16)][4]
// per phase: floor(128 sum_k G_f[k] / 256) as (lo, hi), e32 = ceil(eps_f 2^32) + 2, and first | last << 8 non-zero tap | flag << 16
// with flag = 1 when the phase's outputs are exact integers the screen itself reproduces (a single tap 1.0 at gain 1.0: no
// second look); g: the L x Q double taps; shift as
// in llzs_fir_mfma_i16x; eps = the largest per-phase bound (checked only).  The call must
// start on a period boundary (input index % M == 0, output index % L == 0).
extern "C" int llzs_resample_i16x(const short *in, short *out, const short *hist, const signed char *atab, const int *aoff,
                                  const int *bqtab, const double *g, int channels, long n_in, long n_out, long in_pitch,
                                  long out_pitch, int L, int M, int Q, int shift, double gain, double eps, void *stream)
{
    ri_shape sh;
    int waves;
    size_t lds;
    if (!in || !out || !atab || !aoff || !bqtab || !g || channels <= 0 || channels > 65535 || n_in <= 0 || n_out <= 0 ||
        in_pitch < n_in || out_pitch < n_out || shift < 32 || shift > 46 || !(eps > 0.0) || !(eps < 0.0625) ||
        !llzs_resample_i16x_fits(L, M, Q) || !ri_make_shape(L, M, Q, n_out, channels, &sh, &waves, &lds)) {
        llzs_set_error("resample_i16x: bad arguments (channels=%d L=%d M=%d Q=%d shift=%d eps=%g)", channels, L, M, Q, shift, eps);
        return LLZ_ERR_ARG;
    }
    sh.rs = shift - 40;
    sh.gain = gain;
    const int ks = ri_ksteps(L, M, Q);
    const bool neg = sh.rs < 0, reload = sh.nt > waves;
    const dim3 grid((unsigned)((sh.spans + sh.spans_per_wg - 1) / sh.spans_per_wg), (unsigned)channels), block(64 * waves);
#define RI_GO3(K, N, R)                                                                                               \
    do {                                                                                                              \
        if (lds > 64 * 1024)                                                                                          \
            LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resample_i8x<K, N, R>),                \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                 \
        hipLaunchKernelGGL((k_resample_i8x<K, N, R>), grid, block, lds, as_stream(stream), in, out, hist, atab, aoff, \
                           bqtab, g, n_in, n_out, in_pitch, out_pitch, sh);                                           \
    } while (0)
#define RI_GO2(K, N) do { if (reload) RI_GO3(K, N, true); else RI_GO3(K, N, false); } while (0)
#define RI_GO(K) do { if (neg) RI_GO2(K, true); else RI_GO2(K, false); } while (0)
    if (ks == 1) RI_GO(1);
    else if (ks == 2) RI_GO(2);
    else if (ks == 3) RI_GO(3);
    else RI_GO(4);
#undef RI_GO
#undef RI_GO2
#undef RI_GO3
    LLZ_LAUNCH_CHECK("k_resample_i8x");
    return LLZ_OK;
}
