// fir_mfma.hip -- K1m: time-domain FIR / decimating FIR on the fp32 matrix cores of gfx950.
//
//   y[c][i] = gain * sum_{k<T} taps[k] * x[c][i*M - k]        (M = 1: llz_fir.c:411-426 as driven by :570-580;
//                                                              M > 1: llz_resample.c:583-603 with L = 1)
//
// Why matrix cores for a filter: at 126..270 flop per 8..5.3 bytes the time-domain form is bound by the fp32 VALU,
// not by HBM (fir_td.hip: 44 % of the HBM roofline at 63 taps; resample.hip L=1: 44 %), and v_mfma_f32_16x16x4_f32
// issues the same 64 flop/clk/SIMD as the VALU peak WITHOUT spending issue slots and LDS reads on operand
// shuffling (MI355X_MICROARCH.md: an f32 GEMM runs 122 TF on it against 52 TF on v_pk_fma_f32).  The products and the
// f32 accumulation are exact f32 (one rounding per product-add), so the error budget is that of fir_td.hip.
//
// Mapping.  A wave computes a 16 x 16 tile D[m][n] = output (16*seg(n) + m): column n is one of 16 consecutive
// 16-output segments of a channel, row m the output inside the segment.  With the segment's input window
// w_n[t] = x[16*seg(n)*M - tpad + t]  (t = 0 .. tpad + 15*M),  D = A * B where
//     A[m][t] = gain * taps[m*M + tpad - t]   (zero outside 0..T-1)   -- a banded Toeplitz block, the same for every tile
//     B[t][n] = w_n[t]
// and t is walked four at a time by v_mfma_f32_16x16x4_f32 (lane l supplies A[l&15][k] and B[k][l&15] for ONE k of the
// step).  Which four t values form a step is free as long as A and B agree, so inside each 16-block of t the steps
// are (h, j) = {0,1} x {0,1} with lane group kq = l>>4 supplying t = 8h + 2kq + j: a lane's two B values of a pair
// of steps are adjacent in LDS and come in with one ds_read_b64, its four A values of a block with one ds_read_b128.
// Of the 16 x (tpad + 15M + 1) entries of A, 16*T are non-zero: 75 % at T=134, M=3; 79 % at T=63, M=1.
//
// Data movement.  Persistent workgroups (4 waves) walk tiles of 4*NACC*256 consecutive outputs of one channel.  The
// inputs of a tile (halo tpad) are staged in LDS once with 16-byte coalesced loads; the loads of the NEXT tile are
// issued into registers before the MFMA phase of the current one and written to LDS after it, so HBM latency hides
// behind the matrix work.  The A table [blocks][64 lanes][4] is built once per workgroup.  Each wave runs NACC
// independent accumulators (dependent-accumulator latency 40 cycles > 32-cycle issue).  The LDS image is padded by P
// floats per 16 so that the 16 segment bases, 16*M floats apart, spread over the 64 banks of ds_read_b64 (conflict-free
// for odd M and M = 2 mod 4).  Results leave as one 16-byte non-temporal store per lane and accumulator: a wave writes
// 1 KB contiguous.
#include "common.hpp"
#include <stdlib.h>

#ifndef LLZ_MF_DIAG
#define LLZ_MF_DIAG 0       // ablation builds: 1 = no MFMA phase, 2 = no input loads
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int MF_WAVES = 4;
constexpr int MF_THREADS = MF_WAVES * 64;
constexpr int MF_NV_MAX = 16;              // prefetch registers: 16 x 16 B per thread = at most 16 Ki floats per tile

struct mf_shape {
    int T, M;        // taps, decimation
    int tpad;        // multiple of 4 >= T-1: the window of output i starts at x[i*M - tpad]
    int halves;      // half blocks (8 values of t, two MFMA steps each): ceil((tpad + 15*M + 1) / 8)
    int P;           // LDS pad floats per 16 samples (even)
    int total;       // logical samples staged per tile (multiple of 4)
    int tiles_per_ch;
};

__device__ __forceinline__ int mf_phys(int p, int P) { return p + P * (p >> 4); }

template <int NACC, int MF_NV>
__global__ void __launch_bounds__(MF_THREADS)
k_fir_mfma_f32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
               const float *__restrict__ taps, long n_in, long n_out, long in_pitch, long out_pitch, float gain,
               mf_shape sh, long ntiles)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TILE_OUT = MF_WAVES * NACC * 256;
    const int blocks = (sh.halves + 1) >> 1;
    float *atab = lds;                                   // [blocks][64][4]
    float *xs = lds + blocks * 256;                      // padded input image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int P = sh.P, total = sh.total, last4 = sh.total - 4;
    const bool aligned_in = (in_pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const bool aligned_out = (out_pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;

    // A table: entry (block g, lane, e = 2h + j) = gain * taps[m*M + tpad - t], t = 16g + 8h + 2kq + j
    for (int e = tid; e < blocks * 256; e += MF_THREADS) {
        const int g = e >> 8, l = (e >> 2) & 63, hj = e & 3;
        const int t = 16 * g + 8 * (hj >> 1) + 2 * (l >> 4) + (hj & 1);
        const int k = (l & 15) * sh.M + sh.tpad - t;
        atab[e] = (k >= 0 && k < sh.T) ? taps[k] * gain : 0.f;
    }

    auto tile_first = [&](long q, int &c, long &o0) {
        c = (int)(q / sh.tiles_per_ch);
        o0 = (q - (long)c * sh.tiles_per_ch) * TILE_OUT;
        return o0 * sh.M - sh.tpad;                      // x index of logical sample 0
    };
    auto is_interior = [&](long first) { return aligned_in && first >= 0 && first + total <= n_in; };

    f32x4 v[MF_NV];
    auto prefetch = [&](long q) {                        // uniform result: true when the tile is in registers
        int c; long o0;
        const long first = tile_first(q, c, o0);
        if (LLZ_MF_DIAG == 2 || !is_interior(first)) return false;
        const float *src = in + (size_t)c * in_pitch + first;
#pragma unroll
        for (int j = 0; j < MF_NV; j++) {
            if (j * MF_THREADS * 4 < total) {            // uniform
                int p = (j * MF_THREADS + tid) * 4;
                p = p < last4 ? p : last4;               // clamped, unconditional
                v[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + p));
            }
        }
        return true;
    };

    long q = blockIdx.x;
    bool have = q < ntiles && prefetch(q);
    for (; q < ntiles; q += gridDim.x) {
        int c; long o0;
        const long first = tile_first(q, c, o0);
        __syncthreads();                                 // the previous tile's MFMA phase has finished reading xs
        if (have) {
#pragma unroll
            for (int j = 0; j < MF_NV; j++) {
                if (j * MF_THREADS * 4 < total) {
                    const int p = (j * MF_THREADS + tid) * 4;
                    if (p < total) *reinterpret_cast<f32x2 *>(&xs[mf_phys(p, P)]) = v[j].xy;
                    if (p < total) *reinterpret_cast<f32x2 *>(&xs[mf_phys(p, P) + 2]) = v[j].zw;
                }
            }
        } else if (LLZ_MF_DIAG != 2) {                   // edge tile: history in front, zeros behind
            const float *row = in + (size_t)c * in_pitch;
            for (int base = 0; base < total; base += 8 * MF_THREADS) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {            // clamped, unconditional: eight loads in flight
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
                    if (p < total) xs[mf_phys(p, P)] = (idx >= 0 && idx < n_in) ? x[j] : 0.f;
                }
            }
            if (first < 0 && hist) {                     // same p -> thread mapping as above: ordered by program order
                const float *hrow = hist + (size_t)c * (sh.T - 1);
                for (int p = tid; p < sh.tpad; p += MF_THREADS) {
                    const long idx = first + p;
                    if (idx < 0 && idx >= -(long)(sh.T - 1)) xs[mf_phys(p, P)] = hrow[sh.T - 1 + idx];
                }
            }
        }
        __syncthreads();
        have = q + gridDim.x < ntiles && prefetch(q + gridDim.x);

        f32x4 acc[NACC];
        const float *bp[NACC];
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int s = ((wave * NACC + a) * 16 + n) * 16 * sh.M;      // segment base, multiple of 16
            bp[a] = xs + s + P * (s >> 4) + 2 * kq;
        }
        const float *ap = atab + lane * 4;
        const int full = sh.halves >> 1;
        if (LLZ_MF_DIAG != 1) {
            // two register sets, filled alternately one block ahead of the MFMAs that consume them
            f32x4 a0, a1;
            f32x2 b0[NACC][2], b1[NACC][2];
            const int bstride = 16 + P;
            auto fetch = [&](int g, f32x4 &ad, f32x2 (&bd)[NACC][2]) {
                ad = *reinterpret_cast<const f32x4 *>(ap + g * 256);
#pragma unroll
                for (int a = 0; a < NACC; a++) {
                    bd[a][0] = *reinterpret_cast<const f32x2 *>(bp[a] + g * bstride);
                    bd[a][1] = *reinterpret_cast<const f32x2 *>(bp[a] + g * bstride + 8);
                }
                __builtin_amdgcn_sched_barrier(0);       // keep the reads ahead of the MFMAs that follow
            };
            auto mac = [&](const f32x4 &ad, const f32x2 (&bd)[NACC][2]) {
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int a = 0; a < NACC; a++)
                        acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(ad[e], bd[a][e >> 1][e & 1], acc[a], 0, 0, 0);
            };
            int g = 0;
            if (full > 0) fetch(0, a0, b0);
            for (; g + 2 <= full; g += 2) {
                fetch(g + 1, a1, b1);
                mac(a0, b0);
                if (g + 2 < full) fetch(g + 2, a0, b0);
                mac(a1, b1);
            }
            if (g < full) mac(a0, b0);
            ap += full * 256;
#pragma unroll
            for (int a = 0; a < NACC; a++) bp[a] += full * bstride;
            if (sh.halves & 1) {                         // last half block: two steps
                const f32x2 a2 = *reinterpret_cast<const f32x2 *>(ap);
#pragma unroll
                for (int e = 0; e < 2; e++)
#pragma unroll
                    for (int a = 0; a < NACC; a++) {
                        const f32x2 b2 = *reinterpret_cast<const f32x2 *>(bp[a]);
                        acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[e], b2[e], acc[a], 0, 0, 0);
                    }
            }
        }

        float *orow = out + (size_t)c * out_pitch;
        const bool whole = aligned_out && o0 + TILE_OUT <= n_out;
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            const long o = o0 + ((wave * NACC + a) * 16 + n) * 16 + 4 * kq;   // D rows 4*kq .. 4*kq+3 of column n
            if (whole) {
                __builtin_nontemporal_store(acc[a], reinterpret_cast<f32x4 *>(orow + o));
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (o + j < n_out) orow[o + j] = acc[a][j];
            }
        }
    }
}

bool mf_make_shape(int T, int M, int nacc, long n_out, mf_shape *sh, size_t *lds_bytes)
{
    sh->T = T;
    sh->M = M;
    sh->tpad = (T - 1 + 3) & ~3;
    sh->halves = (sh->tpad + 15 * M + 1 + 7) / 8;
    sh->P = (M & 1) ? 4 : 2;
    const int tile_out = MF_WAVES * nacc * 256;
    sh->total = ((tile_out - 16) * M + 8 * sh->halves + 3) & ~3;
    sh->tiles_per_ch = (int)((n_out + tile_out - 1) / tile_out);
    const size_t image = (size_t)sh->total + (size_t)sh->P * (sh->total >> 4) + 16;
    const size_t blocks = (size_t)(sh->halves + 1) / 2;
    *lds_bytes = (blocks * 256 + image) * sizeof(float);
    return *lds_bytes <= 160 * 1024 && sh->total <= MF_NV_MAX * MF_THREADS * 4;
}

int mf_pick_nacc(int T, int M)
{
    mf_shape sh;
    size_t bytes;
    if (const char *e = getenv("LLZ_MFMA_NACC")) {                   // A/B runs
        const int v = atoi(e);
        if ((v == 1 || v == 2 || v == 4) && mf_make_shape(T, M, v, 1, &sh, &bytes)) return v;
    }
    // two workgroups per CU when the image allows it (LDS write phase of one overlaps the MFMA phase of the other)
    if (mf_make_shape(T, M, 4, 1, &sh, &bytes) && bytes <= 78 * 1024) return 4;
    if (mf_make_shape(T, M, 2, 1, &sh, &bytes) && bytes <= 78 * 1024) return 2;
    if (mf_make_shape(T, M, 1, 1, &sh, &bytes)) return 1;
    return 0;
}

template <int NACC, int NV>
int mf_launch(const float *in, float *out, const float *hist, const float *taps, int channels, long n_in, long n_out,
              long in_pitch, long out_pitch, int T, int M, float gain, void *stream)
{
    mf_shape sh;
    size_t lds_bytes;
    mf_make_shape(T, M, NACC, n_out, &sh, &lds_bytes);
    if (lds_bytes > 64 * 1024)
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir_mfma_f32<NACC, NV>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const long ntiles = (long)sh.tiles_per_ch * channels;
    int cus = 256;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    int per_cu = (int)((160 * 1024) / lds_bytes);
    if (per_cu > 4) per_cu = 4;
    if (const char *e = getenv("LLZ_MFMA_WG_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 8) per_cu = v; }
    long grid = (long)cus * per_cu;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((k_fir_mfma_f32<NACC, NV>), dim3((unsigned)grid), dim3(MF_THREADS), lds_bytes, as_stream(stream), in, out,
                       hist, taps, n_in, n_out, in_pitch, out_pitch, gain, sh, ntiles);
    LLZ_LAUNCH_CHECK("k_fir_mfma_f32");
    return LLZ_OK;
}

} // namespace

extern "C" int llzs_fir_mfma_f32_fits(int T, int M)
{
    return T >= 1 && M >= 1 && mf_pick_nacc(T, M) > 0;
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
    const int nacc = mf_pick_nacc(T, M);
    mf_shape sh;
    size_t bytes;
    if (nacc) mf_make_shape(T, M, nacc, n_out, &sh, &bytes);
    const bool small = nacc && sh.total <= 6 * MF_THREADS * 4;      // fewer prefetch registers -> one more wave per SIMD
#define MF_GO(A, V) return mf_launch<A, V>(in, out, hist, taps, channels, n_in, n_out, in_pitch, out_pitch, T, M, gain, stream)
    switch (nacc) {
    case 4: if (small) MF_GO(4, 6); else MF_GO(4, 16);
    case 2: if (small) MF_GO(2, 6); else MF_GO(2, 16);
    case 1: if (small) MF_GO(1, 6); else MF_GO(1, 16);
#undef MF_GO
    default:
        llzs_set_error("fir_mfma_f32: %d taps at decimation %d do not fit the LDS image", T, M);
        return LLZ_ERR_RANGE;
    }
}
