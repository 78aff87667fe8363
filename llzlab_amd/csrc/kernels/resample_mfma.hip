// resample_mfma.hip -- K3g: rational L/M resampling with large L and M (147:160, 160:147 ...) on the fp32 matrix cores.
//
//   reference (llz_resample.c:583-603):   y[i] = gain * sum_{k<Q} x[(i M) / L - k] * g[i mod L][k]
//
// An L:M resampler is L decimators by M interleaved at the output: with i = L m + f (period m, phase f < L)
//       y[L m + f] = sum_k x[M m + c_f - k] * g_f[k],        c_f = floor(f M / L).
// For ONE period m every phase reads inside the same window of M + Q input samples, and the taps do not depend on m:
//       Y[f][m] = sum_u W[f][u] * X[u][m],    W[f][u] = g_f[c_f - u]  (zero outside 0 <= c_f - u < Q),   X[u][m] = x[M m + u]
// -- a product of a FIXED banded matrix W (L x (M + Q)) with the signal laid out as overlapping columns.  The VALU form of
// this (k_resample_f32_lds, resample.hip) needs one LDS read per multiply-add and stops at 12-13 % of the HBM roofline;
// v_mfma_f32_16x16x4_f32 takes one LDS read per 16 multiply-adds and keeps fp32 products and sums (no split needed).
//
// Mapping.  A wave owns a tile of 16 phases x 16 periods: D[r][n] = output (phase 16 T + r, period m_n).  The band of 16
// consecutive phases is c_{16T+15} - c_{16T} + Q wide (64 samples at 147:160, Q = 47), walked 4 samples per MFMA starting
// at u0 = c_{16T} - (Q-1):   A[r][t] = gain * g_{16T+r}[c_{16T+r} - u0 - t],   B[t][n] = x[M m_n + u0 + t].
// A is built once on the host ([phase tile][step][64 lanes]) and lives in the wave's registers while it sweeps a phase
// tile; the signal span of a workgroup (64 periods + Q-1 samples of history) is staged once in LDS, one float per
// sample, with the period stride made odd (one pad float per period when M is even) so that the 16 columns of a B read
// fall in 16 different banks.  A workgroup = 4 waves = 4 period tiles; every wave sweeps all phase tiles, so the work
// is balanced for any L, and a wave's stores of one tile are 16 runs of 16 consecutive floats that the next phase tile
// extends (whole lines form in L2 before they leave for HBM).
#include "common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RM_MAX_WAVES = 4;                   // a workgroup is 1..4 waves = 16..64 periods (fewer for long periods)
constexpr int RM_CHUNK = 2;                       // phase tiles per output run: 32 consecutive outputs of a period
constexpr int RM_SROW = 16 * RM_CHUNK + 4;        // floats per row of the per-wave output buffer (+4: 16-byte writes of the
                                                  // 16 columns land in different banks)

struct rm_shape {
    int L, M, Q;
    int ntiles;          // phase tiles: ceil(L / 16)
    int pad;             // pad floats per period: column stride M + pad = 2 mod 32, so that the 32 lanes of a B read
                         // (16 columns x 2 consecutive samples) fall in 32 different banks
    int H;               // history samples in front of the span: Q - 1
    int img;             // floats of the signal image (with padding)
    unsigned m_magic;    // ceil(2^32 / M): e / M for e < 2^32 / M by multiply-high
};

template <int KS>
__global__ void __launch_bounds__(RM_MAX_WAVES * 64)
k_resample_mfma_f32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
                    const float *__restrict__ atab /* [ntiles][KS][64] */, const int *__restrict__ c0tab /* [ntiles] */,
                    long n_in, long n_out, long in_pitch, long out_pitch, rm_shape sh)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves = blockDim.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int c = blockIdx.y;
    const int periods = 16 * waves;
    const long m0 = (long)blockIdx.x * periods;                     // first period of this workgroup
    const long first = m0 * sh.M - sh.H;                            // input index of image element 0
    // (4 KS more than the span: the zero-padded tail of a band reads there; the values only have to be finite)
    const int count = periods * sh.M + sh.H + 4 * KS;
    const float *row = in + (size_t)c * in_pitch;
    auto pos_of = [&](int e) { return e + sh.pad * (int)__umulhi((unsigned)e, sh.m_magic); };
    // 16 loads in flight per lane (a one-load-per-iteration loop waits out the full memory latency 40 times per span)
    const bool interior = first >= 0 && first + count <= n_in;
    const float *hrow = hist ? hist + (size_t)c * (sh.Q - 1) : nullptr;
    for (int base = 0; base < count; base += 16 * (int)blockDim.x) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int p = base + j * (int)blockDim.x + tid;
            const long idx = first + p;
            v[j] = 0.f;
            if (interior) {
                if (p < count) v[j] = __builtin_nontemporal_load(&row[idx]);
            } else if (p < count) {
                if (idx >= 0) {
                    if (idx < n_in) v[j] = row[idx];
                } else if (hrow && idx >= -(long)(sh.Q - 1)) {
                    v[j] = hrow[(sh.Q - 1) + idx];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int p = base + j * (int)blockDim.x + tid;
            if (p < count) xs[pos_of(p)] = v[j];
        }
    }
    __syncthreads();

    // this wave's 16 periods: column n is period m0 + 16 wave + n; its window for a phase tile starts at image element
    // (16 wave + n) M + c0(tile)   [history H and band start c0 - (Q-1) cancel]
    const long mw = m0 + 16 * wave;                                  // the wave's first period
    float *orow = out + (size_t)c * out_pitch;
    // results leave through a per-wave LDS buffer [16 periods][64 phases]: a wave instruction then stores 64 CONSECUTIVE
    // outputs of one period (stored straight from the accumulators a wave instruction would touch 64 separate dwords)
    float *stage = xs + sh.img + wave * (16 * RM_SROW);
    // Window sample d = c0 + kq + 4 s of column n sits at image position colpos + d + pad * (d / M); d < M + 4 KS, so with
    // M >= 4 KS the quotient is 0 or 1: two base addresses and one select per step instead of a division
    const int colpos = (16 * wave + n) * (sh.M + sh.pad);
    auto load_tile = [&](int t, float (&av)[KS], float (&bv)[KS]) {
        const float *ap = atab + ((size_t)t * KS) * 64 + lane;
        const int d0 = c0tab[t] + kq;
        const float *lo = xs + colpos + d0, *hi = lo + sh.pad;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            av[s] = ap[s * 64];
            bv[s] = (d0 + 4 * s >= sh.M ? hi : lo)[4 * s];
        }
    };
    auto finish_tile = [&](int t, const f32x4 &acc) {
        // D[4 kq + j][n] = output (phase 16 t + 4 kq + j, period mw + n) -> stage[n][16 (t % CHUNK) + 4 kq + j]
        *reinterpret_cast<f32x4 *>(&stage[n * RM_SROW + 16 * (t % RM_CHUNK) + 4 * kq]) = acc;
        if ((t % RM_CHUNK) == RM_CHUNK - 1 || t == sh.ntiles - 1) {
            const int f0 = 16 * (t - t % RM_CHUNK);                  // first phase of the run
            const int run = min(16 * RM_CHUNK, sh.L - f0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's own LDS writes have landed
            // two periods per instruction: lanes 0..31 one run of 32 outputs, lanes 32..63 the next period's
            const int half = lane >> 5, l5 = lane & 31;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const long o = (mw + r + half) * sh.L + f0 + l5;
                if (l5 < run && o < n_out) __builtin_nontemporal_store(stage[(r + half) * RM_SROW + l5], &orow[o]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // reads done before the next chunk overwrites the buffer
        }
    };
    auto products = [&](const float (&av)[KS], const float (&bv)[KS]) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
        return acc;
    };
    // two register sets in turn: a tile's taps and samples arrive during the previous tile's products
    float a0[KS], b0[KS], a1[KS], b1[KS];
    load_tile(0, a0, b0);
#pragma unroll 1
    for (int t = 0; t < sh.ntiles; t += 2) {
        if (t + 1 < sh.ntiles) load_tile(t + 1, a1, b1);
        finish_tile(t, products(a0, b0));
        if (t + 1 < sh.ntiles) {
            if (t + 2 < sh.ntiles) load_tile(t + 2, a0, b0);
            finish_tile(t + 1, products(a1, b1));
        }
    }
}

} // namespace

extern "C" int llzs_resample_mfma_f32_ksteps(int L, int M, int Q)
{
    // widest band over the phase tiles, in steps of 4 samples
    int widest = 0;
    for (int f0 = 0; f0 < L; f0 += 16) {
        const int f1 = f0 + 15 < L ? f0 + 15 : L - 1;
        const int w = (int)(((long)f1 * M) / L - ((long)f0 * M) / L) + Q;
        if (w > widest) widest = w;
    }
    return (widest + 3) / 4;
}

static int rm_pad(int M) { return ((2 - M) % 32 + 32) % 32; }

// waves per workgroup: as many period tiles (16 periods each) as fit ~48 KB of signal image, at most 4
static int rm_waves(int M)
{
    int w = (48 * 1024) / (16 * (M + rm_pad(M)) * (int)sizeof(float));
    return w < 1 ? 1 : (w > RM_MAX_WAVES ? RM_MAX_WAVES : w);
}

static size_t rm_image_floats(int M, int Q, int waves, int kst)
{
    const size_t span = (size_t)16 * waves * M + (Q - 1) + 4 * kst;
    return span + (size_t)rm_pad(M) * (span / M) + 8;
}

extern "C" int llzs_resample_mfma_f32_table_steps(int L, int M, int Q)
{
    const int ks = llzs_resample_mfma_f32_ksteps(L, M, Q);
    return ks <= 8 ? 8 : ks <= 16 ? 16 : ks <= 24 ? 24 : 32;
}

extern "C" int llzs_resample_mfma_f32_fits(int L, int M, int Q)
{
    // small L: the register-window kernel; short periods (M < a band's 4 KS samples) give a workgroup too little to do per
    // staged span: the LDS form is faster there (8:7: 1.8 against 3.8 ms)
    if (L < 5 || M < 1 || Q < 1) return 0;
    if (llzs_resample_mfma_f32_ksteps(L, M, Q) > 32) return 0;
    const int waves = rm_waves(M), kst = llzs_resample_mfma_f32_table_steps(L, M, Q);
    const size_t lds = (rm_image_floats(M, Q, waves, kst) + 4 + (size_t)waves * 16 * RM_SROW) * sizeof(float);
    return M >= 4 * kst && lds <= 160 * 1024 && (long)16 * waves * M + Q + 128 < (long)(0x100000000ull / (unsigned)M);
}

// atab: [ceil(L/16)][steps][64] floats (steps = llzs_resample_mfma_f32_table_steps), gain folded in; c0tab: [ceil(L/16)]
// ints = floor(16 t M / L).  The call must start on a period boundary.
extern "C" int llzs_resample_mfma_f32(const float *in, float *out, const float *hist, const float *atab, const int *c0tab,
                                      int channels, long n_in, long n_out, long in_pitch, long out_pitch, int L, int M, int Q,
                                      void *stream)
{
    if (!in || !out || !atab || !c0tab || channels <= 0 || channels > 65535 || n_in <= 0 || n_out <= 0 ||
        in_pitch < n_in || out_pitch < n_out || !llzs_resample_mfma_f32_fits(L, M, Q)) {
        llzs_set_error("resample_mfma_f32: bad arguments (channels=%d L=%d M=%d Q=%d)", channels, L, M, Q);
        return LLZ_ERR_ARG;
    }
    const int waves = rm_waves(M), kst = llzs_resample_mfma_f32_table_steps(L, M, Q);
    rm_shape sh;
    sh.L = L; sh.M = M; sh.Q = Q;
    sh.ntiles = (L + 15) / 16;
    sh.pad = rm_pad(M);
    sh.H = Q - 1;
    sh.img = (int)rm_image_floats(M, Q, waves, kst);
    sh.img = (sh.img + 3) & ~3;                                              // 16-byte aligned staging rows behind it
    sh.m_magic = (unsigned)((0x100000000ull + (unsigned)M - 1) / (unsigned)M);
    const long periods = (n_out + L - 1) / L;
    const dim3 grid((unsigned)((periods + 16 * waves - 1) / (16 * waves)), (unsigned)channels), block(64 * waves);
    const size_t lds = ((size_t)sh.img + (size_t)waves * 16 * RM_SROW) * sizeof(float);
#define RM_GO(K)                                                                                                     \
    do {                                                                                                             \
        if (lds > 64 * 1024)                                                                                         \
            LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_resample_mfma_f32<K>),                \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                \
        hipLaunchKernelGGL(k_resample_mfma_f32<K>, grid, block, lds, as_stream(stream), in, out, hist, atab, c0tab,  \
                           n_in, n_out, in_pitch, out_pitch, sh);                                                    \
    } while (0)
    if (kst == 8) RM_GO(8);
    else if (kst == 16) RM_GO(16);
    else if (kst == 24) RM_GO(24);
    else RM_GO(32);
#undef RM_GO
    LLZ_LAUNCH_CHECK("k_resample_mfma_f32");
    return LLZ_OK;
}
