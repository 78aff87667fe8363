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
// period n).  The band of a phase tile starts at window position a_t (aligned down to 16 bytes) and is walked 64 samples per
// step; A (the tile's tap digits, [step][plane] in operand order, built on the host) stays in the wave's registers while it
// walks the period tiles of a span.  The LDS image holds one COLUMN per period: the period's whole window as two byte planes
// (low digit (x & 255) - 128, high digit x >> 8), column stride an odd number of 16-byte chunks, so that a B operand is one
// aligned ds_read_b128 at column base + a_t + 64 s + 16 chunk(kq), conflict-free (screen_i8.hpp).  Neighbouring columns
// overlap (a sample is stored in every column whose window holds it: ~1.7 copies at 147:160).  Results go to an LDS image of
// the output in memory order and leave in 8-byte pieces; the next span's samples are requested one span ahead into registers.
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
    int clen;               // window positions a column holds (bytes per plane actually used)
    int cstride;            // bytes between columns of a plane: >= clen, an odd multiple of 16
    int pt, P;              // period tiles per span, periods per span = 16 pt
    int count;              // samples a span touches: M (P - 1) + clen
    int plane;              // bytes per plane: P cstride
    int ngroups;            // 16-byte groups per span (count + 7 alignment slack, / 8, rounded up)
    unsigned m_magic;       // ceil(2^32 / M)
    long spans;             // spans of a channel
    int spans_per_wg;
    int rs;                 // shift - 40
    unsigned e32;
    double gain;
};

// the reference's loop for one output from the column's planes: sample at window position p is 256 hi[p] + lo[p] + 128
__device__ __forceinline__ short ri_exact(const signed char *hi, const signed char *lo, int p0, const double *__restrict__ g, int Q,
                                          double gain)
{
#pragma clang fp contract(off)
    double y = 0.0;
#pragma unroll 1
    for (int k = 0; k < Q; k++) {
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
               const int *__restrict__ bqtab /* [16 nt][2] */, const double *__restrict__ g /* [L][Q] */, long n_in, long n_out,
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
    const bool m8 = (sh.M & 7) == 0;

    // a span's samples: absolute indices [M m0 - Hp, M m0 - Hp + count), requested in 16-byte groups from the 8-aligned index
    // at or below the start
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
        const int lead = (int)(s0 - (s0 >= 0 ? (s0 & ~7L) : -((-s0 + 7) & ~7L)));        // samples of the first group in front of s0
        // ---- the requested groups into the column image (the previous span's readers passed the barrier at its end) ----
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
                const int e0 = 8 * gi - lead;                                   // span-relative index of the group's first sample
                if (m8) {
                    // M, Hp and the group starts are multiples of 8: a group stays together in every column that holds it
                    if (e0 >= 0 && e0 < sh.count) {
                        int q = (int)__umulhi((unsigned)e0, sh.m_magic);
                        int pos = e0 - q * sh.M;
                        for (; q >= 0 && pos < sh.clen; q--, pos += sh.M)
                            if (q < P) {
                                *reinterpret_cast<u32x2 *>(&xs_lo[q * sh.cstride + pos]) = lo;
                                *reinterpret_cast<u32x2 *>(&xs_hi[q * sh.cstride + pos]) = hi;
                            }
                    }
                } else {
#pragma unroll 1
                    for (int e = 0; e < 8; e++) {
                        const int ee = e0 + e;
                        if (ee >= 0 && ee < sh.count) {
                            const int sft = 8 * (e & 3);
                            const signed char bl = (signed char)((e < 4 ? lo[0] : lo[1]) >> sft);
                            const signed char bh = (signed char)((e < 4 ? hi[0] : hi[1]) >> sft);
                            int q = (int)__umulhi((unsigned)ee, sh.m_magic);
                            int pos = ee - q * sh.M;
                            for (; q >= 0 && pos < sh.clen; q--, pos += sh.M)
                                if (q < P) {
                                    xs_lo[q * sh.cstride + pos] = bl;
                                    xs_hi[q * sh.cstride + pos] = bh;
                                }
                        }
                    }
                }
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
#pragma unroll
            for (int j = 0; j < 4; j++) {
                bql[j] = bqtab[2 * (f0 + j)];
                bqh[j] = bqtab[2 * (f0 + j) + 1];
            }
#pragma unroll 1
            for (int p = 0; p < sh.pt; p++) {
                const int col = 16 * p + col16;
                const signed char *bp = xs_lo + col * sh.cstride + a_t + 16 * ck;
                scr_i32x4 acc[5];
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    const scr_i32x4 b_lo = *reinterpret_cast<const scr_i32x4 *>(bp + 64 * s);
                    const scr_i32x4 b_hi = *reinterpret_cast<const scr_i32x4 *>(bp + sh.plane + 64 * s);
                    if (s == 0) scr_step<true>(acc, ad[s], b_lo, b_hi);
                    else scr_step<false>(acc, ad[s], b_lo, b_hi);
                }
                int res[4];
                unsigned mine = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    bool unsure;
                    res[j] = scr_decide<NEG>(acc[0][j], acc[1][j], acc[2][j], acc[3][j], acc[4][j], bql[j], bqh[j], sh.rs, sh.e32,
                                             unsure);
                    mine = mine + mine + (unsure ? 1u : 0u);                      // slot j at bit 3 - j
                }
                if (__ballot(mine != 0) != 0) {
#pragma unroll 1
                    while (mine != 0) {
                        // an integer within eps of the value: the reference's own arithmetic decides, in the lane that found it
                        const int j = 3 - __builtin_ctz(mine);
                        mine &= mine - 1;
                        const int f = f0 + j;
                        int cur = 0;
#pragma unroll
                        for (int u = 0; u < 4; u++) cur = j == u ? res[u] : cur;
                        // (a value far outside the clamp range needs no second look, nor does a row or period that does not exist)
                        if (f < sh.L && col < periods_left && scr_in_reach(cur)) {
                            const int cf = (int)(((long)f * sh.M) / sh.L);
                            const int r = ri_exact(xs_hi + col * sh.cstride, xs_lo + col * sh.cstride, cf + sh.Hp, g + (size_t)f * sh.Q,
                                                   sh.Q, sh.gain);
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

int ri_ksteps(int L, int M, int Q, int *clen_out)
{
    const int Hp = (Q - 1 + 7) & ~7, nt = (L + 15) / 16;
    int ks = 1, amax = 0;
    for (int t = 0; t < nt; t++) {
        const int flast = 16 * t + 15 < L ? 16 * t + 15 : L - 1;
        const int a = (int)(((long)16 * t * M) / L + Hp - (Q - 1)) & ~15;
        const int top = (int)(((long)flast * M) / L) + Hp;                    // highest position the tile reads (k = 0)
        const int need = (top - a + 1 + 63) / 64;
        if (need > ks) ks = need;
        if (a > amax) amax = a;
    }
    if (clen_out) *clen_out = amax + 64 * ks;
    return ks;
}

} // namespace

extern "C" int llzs_resample_i16x_ksteps(int L, int M, int Q) { return ri_ksteps(L, M, Q, nullptr); }

// geometry of the launch (also used by the host to size nothing: all tables depend on L, M, Q only)
static bool ri_make_shape(int L, int M, int Q, long n_out, int channels, ri_shape *sh, int *waves, size_t *lds)
{
    sh->L = L; sh->M = M; sh->Q = Q;
    sh->nt = (L + 15) / 16;
    sh->Hp = (Q - 1 + 7) & ~7;
    const int ks = ri_ksteps(L, M, Q, &sh->clen);
    if (ks > 4) return false;
    int chunks = (sh->clen + 15) / 16;
    if ((chunks & 1) == 0) chunks++;
    sh->cstride = 16 * chunks;
    // waves: the phase tiles dealt evenly over at most 8 waves (147 phases: 10 tiles -> 5 waves x 2 tiles)
    const int rounds = (sh->nt + 7) / 8;
    int w = (sh->nt + rounds - 1) / rounds;
    if (w < 2) w = 2;
    *waves = w;
    for (int pt = 4; pt >= 1; pt--) {
        sh->pt = pt; sh->P = 16 * pt;
        sh->plane = sh->P * sh->cstride;
        sh->count = M * (sh->P - 1) + sh->clen;
        sh->ngroups = (sh->count + 7 + 7) / 8;
        *lds = 2 * (size_t)sh->plane + sizeof(short) * (size_t)sh->P * L;
        while (*waves < 8 && sh->ngroups > RI_NG * 64 * *waves) (*waves)++;
        if (*lds <= 52 * 1024 && sh->ngroups <= RI_NG * 64 * *waves) break;
        if (pt == 1) return *lds <= 160 * 1024 && sh->ngroups <= RI_NG * 64 * *waves;
    }
    sh->m_magic = (unsigned)((0x100000000ull + (unsigned)M - 1) / (unsigned)M);
    if ((unsigned long long)sh->count >= 0x100000000ull / (unsigned)M) return false;
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

// atab: [ceil(L/16)][ksteps][5][64][16] tap digits in operand order; aoff: [ceil(L/16)] band starts; bqtab: [16 ceil(L/16)][2]
// floor(128 sum_k G_f[k] / 256) as (lo, hi); g: the L x Q double taps; shift / eps as in llzs_fir_mfma_i16x.  The call must
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
    sh.e32 = (unsigned)ceil(ldexp(eps, 32)) + 2u;
    sh.gain = gain;
    const int ks = ri_ksteps(L, M, Q, nullptr);
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
