// fir_td.hip -- K1: time-domain multi-channel FIR for gfx950, plus the exact-order double kernel behind the
// reference's single-channel llz_fir_filter symbol and the history (tail) update.
//
// Replaces the inner loop of reference libllzfilter/llz_fir.c:411-426 (llz_conv) as driven by llz_fir.c:570-580.
//
// Layout: planar [channels][n] float32. One workgroup (256 threads, 4 waves) owns a tile of 2048 consecutive
// outputs of one channel. The tile plus its flt_len halo is staged once in LDS with 16-byte coalesced reads;
// each lane then produces 8 consecutive outputs from a sliding register window that it refills with two
// ds_read_b128 per 8 taps (64 FMAs per 2 LDS reads). Taps are wave-uniform and come through the scalar cache.
// Bound: HBM at 63 taps (126 flop per 8 B), VALU at 257 taps (514 flop per 8 B) -- the long filter goes to
// fir_ols.hip instead.
// Tried and dropped (same-box A/B, 4096 ch x 63 taps): a persistent grid-stride version that prefetches the next tile
// into registers ran 13.1 ms against 9.65 ms for this one-tile-per-workgroup form: the hardware's workgroup dispatch
// overlaps load and compute phases of different tiles better than two extra barriers per tile allow.
#include "common.hpp"

namespace {

constexpr int TD_R = 8;                    // outputs per lane
constexpr int TD_THREADS = 256;
constexpr int TD_TILE = TD_R * TD_THREADS; // 2048 outputs per workgroup

// LDS image: 4 pad floats after every 8 samples, so that lanes 8 samples apart read ds_read_b128 from
// addresses 12 dwords apart: conflict-free within each 16-lane service group (MI355X LDS: 64 banks for b128).
__device__ __forceinline__ int td_phys(int p) { return p + ((p >> 3) << 2); }

__global__ void __launch_bounds__(TD_THREADS)
k_fir_td_f32(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ hist,
             const float *__restrict__ taps, int n, long in_pitch, long out_pitch, int flt_len, int tpad)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int c = blockIdx.y;
    const int tile0 = blockIdx.x * TD_TILE;
    const int tid = threadIdx.x;
    const float *row = in + (size_t)c * in_pitch;
    const float *hrow = hist ? hist + (size_t)c * (flt_len - 1) : nullptr;
    const int halo = tpad;                          // multiple of 16 >= flt_len
    const int total = TD_TILE + halo;               // logical samples staged
    const int first = tile0 - halo;                 // sample index of logical position 0

    const bool interior = (first >= 0) && (tile0 + TD_TILE <= n) && ((in_pitch & 3) == 0) &&
                          ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
    if (interior) {
        for (int p = tid * 4; p < total; p += TD_THREADS * 4) {
            float4 v = *reinterpret_cast<const float4 *>(row + first + p);
            *reinterpret_cast<float4 *>(&lds[td_phys(p)]) = v;
        }
    } else {
        for (int p = tid; p < total; p += TD_THREADS) {
            const int idx = first + p;
            float v = 0.f;
            if (idx >= 0) {
                if (idx < n) v = row[idx];
            } else if (hrow && idx >= -(flt_len - 1)) {
                v = hrow[flt_len - 1 + idx];
            }
            lds[td_phys(p)] = v;
        }
    }
    __syncthreads();

    float acc[TD_R];
#pragma unroll
    for (int r = 0; r < TD_R; r++) acc[r] = 0.f;

    // window registers: `hi` holds logical [base+8, base+16), `lo` holds [base, base+8) with
    // base = halo + tid*8 - kc - 8; tap k = kc+kk meets output r at window slot 8 + r - kk
    const int p0 = halo + tid * TD_R;
    float wa[8], wb[8];
    {
        const float4 a = *reinterpret_cast<const float4 *>(&lds[td_phys(p0)]);
        const float4 b = *reinterpret_cast<const float4 *>(&lds[td_phys(p0 + 4)]);
        wa[0] = a.x; wa[1] = a.y; wa[2] = a.z; wa[3] = a.w;
        wa[4] = b.x; wa[5] = b.y; wa[6] = b.z; wa[7] = b.w;
    }

    auto load8 = [&](float (&w)[8], int p) {
        const float4 a = *reinterpret_cast<const float4 *>(&lds[td_phys(p)]);
        const float4 b = *reinterpret_cast<const float4 *>(&lds[td_phys(p + 4)]);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
        w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    };
    auto mac8 = [&](const float (&lo)[8], const float (&hi)[8], const float *h8) {
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            const float h = h8[kk];
#pragma unroll
            for (int r = 0; r < TD_R; r++) {
                const int slot = 8 + r - kk;            // 1..15
                const float x = slot >= 8 ? hi[slot - 8] : lo[slot];
                acc[r] = __builtin_fmaf(h, x, acc[r]);
            }
        }
    };

    for (int kc = 0; kc < tpad; kc += 16) {
        load8(wb, p0 - kc - 8);
        mac8(wb, wa, taps + kc);
        load8(wa, p0 - kc - 16);
        mac8(wa, wb, taps + kc + 8);
    }

    float *orow = out + (size_t)c * out_pitch;
    const int o0 = tile0 + tid * TD_R;
    if (o0 + TD_R <= n && ((out_pitch & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
        *reinterpret_cast<float4 *>(orow + o0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4 *>(orow + o0 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    } else {
#pragma unroll
        for (int r = 0; r < TD_R; r++)
            if (o0 + r < n) orow[o0 + r] = acc[r];
    }
}

// hist_new[c][j] = sample (n - (T-1) + j) of concat(hist_old, in): one thread per element
__global__ void __launch_bounds__(256)
k_fir_tail_f32(const float *__restrict__ in, const float *__restrict__ hist_old, float *__restrict__ hist_new,
               int n, long in_pitch, int keep)
{
    const int c = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= keep) return;
    const long idx = (long)n - keep + j;            // index into this call's input; negative -> old history
    float v;
    if (idx >= 0) v = in[(size_t)c * in_pitch + idx];
    else v = hist_old[(size_t)c * keep + (keep + idx)];
    hist_new[(size_t)c * keep + j] = v;
}

// Single channel, double. The reference's arithmetic exactly: y = 0; for k ascending: y += h[k]*x[i-k]
// as a rounded multiply followed by a rounded add (x86-64 -O2 has no FMA contraction).
__global__ void __launch_bounds__(256)
k_fir_td_f64_exact(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ hist,
                   const double *__restrict__ taps, int n, int flt_len)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double y = 0.0;
    for (int k = 0; k < flt_len; k++) {
        const int idx = i - k;
        const double x = idx >= 0 ? in[idx] : hist[flt_len - 1 + idx];
        const double prod = taps[k] * x;
        y = y + prod;
    }
    out[i] = y;
}

} // namespace

static size_t td_lds_bytes(int flt_len)
{
    const int tpad = (flt_len + 15) & ~15;
    const int total = TD_TILE + tpad;
    return (size_t)(total + (total >> 3) * 4 + 16) * sizeof(float);
}

// 1 when a filter of flt_len taps fits the kernel's LDS tile (the flush of every FIR handle runs through this kernel)
extern "C" int llzs_fir_td_f32_fits(int flt_len) { return flt_len >= 1 && td_lds_bytes(flt_len) <= 160 * 1024; }

extern "C" int llzs_fir_td_f32(const float *in, float *out, const float *hist, const float *taps_padded,
                               int channels, int n, long in_pitch, long out_pitch, int flt_len, void *stream)
{
    if (!in || !out || !taps_padded || channels <= 0 || n <= 0 || flt_len <= 0 || in_pitch < n ||
        out_pitch < n || channels > 65535) {
        llzs_set_error("fir_td_f32: bad arguments (channels=%d n=%d flt_len=%d)", channels, n, flt_len);
        return LLZ_ERR_ARG;
    }
    const int tpad = (flt_len + 15) & ~15;
    const size_t lds_bytes = td_lds_bytes(flt_len);
    if (lds_bytes > 160 * 1024) {
        llzs_set_error("fir_td_f32: %d taps need %zu B of LDS", flt_len, lds_bytes);
        return LLZ_ERR_RANGE;
    }
    if (lds_bytes > 64 * 1024)
        LLZ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fir_td_f32),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    dim3 grid((unsigned)((n + TD_TILE - 1) / TD_TILE), (unsigned)channels);
    hipLaunchKernelGGL(k_fir_td_f32, grid, dim3(TD_THREADS), lds_bytes, as_stream(stream), in, out, hist,
                       taps_padded, n, in_pitch, out_pitch, flt_len, tpad);
    LLZ_LAUNCH_CHECK("k_fir_td_f32");
    return LLZ_OK;
}

extern "C" int llzs_fir_tail_f32(const float *in, const float *hist_old, float *hist_new, int channels, int n,
                                 long in_pitch, int flt_len, void *stream)
{
    const int keep = flt_len - 1;
    if (keep <= 0) return LLZ_OK;
    if (!in || !hist_old || !hist_new || channels <= 0 || n <= 0 || channels > 65535) {
        llzs_set_error("fir_tail_f32: bad arguments");
        return LLZ_ERR_ARG;
    }
    dim3 grid((unsigned)((keep + 255) / 256), (unsigned)channels);
    hipLaunchKernelGGL(k_fir_tail_f32, grid, dim3(256), 0, as_stream(stream), in, hist_old, hist_new, n,
                       in_pitch, keep);
    LLZ_LAUNCH_CHECK("k_fir_tail_f32");
    return LLZ_OK;
}

extern "C" int llzs_fir_td_f64(const double *in, double *out, const double *hist, const double *taps, int n,
                               int flt_len, void *stream)
{
    if (!in || !out || !hist || !taps || n <= 0 || flt_len <= 0) {
        llzs_set_error("fir_td_f64: bad arguments");
        return LLZ_ERR_ARG;
    }
    hipLaunchKernelGGL(k_fir_td_f64_exact, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       in, out, hist, taps, n, flt_len);
    LLZ_LAUNCH_CHECK("k_fir_td_f64_exact");
    return LLZ_OK;
}
