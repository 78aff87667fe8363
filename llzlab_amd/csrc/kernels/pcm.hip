// pcm.hip -- interleaved int16 <-> planar float32.  Pure data movement: 2 B + 4 B per
// sample-channel, bound by HBM.  A workgroup transposes a tile of TC channels x TS samples (TC*TS <= 4096) through LDS
// so that both sides see contiguous runs: TC int16 per sample on the interleaved side (whole rows when channels <= 64),
// TS floats per channel on the planar side.
#include "common.hpp"

namespace {

constexpr int PCM_THREADS = 256;
constexpr int PCM_TILE = 4096;

struct pcm_tile {
    int tc, ts;              // channels and samples per tile
    unsigned tc_magic, ts_magic;   // ceil(2^32 / tc), ceil(2^32 / ts): e / d for e < 4096 by multiply-high
};

__device__ __forceinline__ int div_magic(int e, unsigned magic, int d)
{
    return d == 1 ? e : (int)__umulhi((unsigned)e, magic);
}

template <bool DEINTERLEAVE>
__global__ void __launch_bounds__(PCM_THREADS)
k_pcm_transpose(const short *__restrict__ ileaved_in, float *__restrict__ planar_out,
                const float *__restrict__ planar_in, short *__restrict__ ileaved_out, int channels, long n,
                float scale, pcm_tile t)
{
    __shared__ float tile[PCM_TILE + PCM_TILE / 32 + 64];            // [tc][ts] with one pad float per 32
    const int tid = threadIdx.x;
    const long i0 = (long)blockIdx.x * t.ts;
    const int c0 = blockIdx.y * t.tc;
    const int tc = min(t.tc, channels - c0);
    const int ts = (int)min((long)t.ts, n - i0);
    const int total = t.tc * t.ts;
    if (DEINTERLEAVE) {
        for (int e = tid; e < total; e += PCM_THREADS) {               // e = s*TC + c: contiguous along channels
            const int s = div_magic(e, t.tc_magic, t.tc), c = e - s * t.tc;
            if (s < ts && c < tc) {
                const int a = c * t.ts + s;
                tile[a + (a >> 5)] = (float)ileaved_in[(i0 + s) * channels + c0 + c] * scale;
            }
        }
        __syncthreads();
        for (int e = tid; e < total; e += PCM_THREADS) {               // e = c*TS + s: contiguous along time
            const int c = div_magic(e, t.ts_magic, t.ts), s = e - c * t.ts;
            if (s < ts && c < tc) planar_out[(size_t)(c0 + c) * n + i0 + s] = tile[e + (e >> 5)];
        }
    } else {
        for (int e = tid; e < total; e += PCM_THREADS) {
            const int c = div_magic(e, t.ts_magic, t.ts), s = e - c * t.ts;
            if (s < ts && c < tc) tile[e + (e >> 5)] = planar_in[(size_t)(c0 + c) * n + i0 + s];
        }
        __syncthreads();
        for (int e = tid; e < total; e += PCM_THREADS) {
            const int s = div_magic(e, t.tc_magic, t.tc), c = e - s * t.tc;
            if (s < ts && c < tc) {
                const int a = c * t.ts + s;
                float y = tile[a + (a >> 5)] * scale;
                y = fminf(fmaxf(y, -32768.f), 32767.f);                // llz_resample.c:596-599
                ileaved_out[(i0 + s) * channels + c0 + c] = (short)(int)y;   // :601, truncation toward zero
            }
        }
    }
}

int pcm_launch(bool deint, const void *in, void *out, int channels, long n, float scale, void *stream)
{
    if (!in || !out || channels < 1 || n < 1) {
        llzs_set_error("pcm (de)interleave: bad arguments");
        return LLZ_ERR_ARG;
    }
    pcm_tile t;
    t.tc = channels < 64 ? channels : 64;
    t.ts = PCM_TILE / t.tc;
    t.tc_magic = (unsigned)((0x100000000ull + (unsigned)t.tc - 1) / (unsigned)t.tc);
    t.ts_magic = (unsigned)((0x100000000ull + (unsigned)t.ts - 1) / (unsigned)t.ts);
    const long bx = (n + t.ts - 1) / t.ts;
    const long by = ((long)channels + t.tc - 1) / t.tc;
    if (by > 65535 || bx > 0x7fffffffL) {
        llzs_set_error("pcm (de)interleave: too many tiles");
        return LLZ_ERR_RANGE;
    }
    dim3 grid((unsigned)bx, (unsigned)by);
    if (deint)
        hipLaunchKernelGGL(k_pcm_transpose<true>, grid, dim3(PCM_THREADS), 0, as_stream(stream), (const short *)in,
                           (float *)out, (const float *)nullptr, (short *)nullptr, channels, n, scale, t);
    else
        hipLaunchKernelGGL(k_pcm_transpose<false>, grid, dim3(PCM_THREADS), 0, as_stream(stream),
                           (const short *)nullptr, (float *)nullptr, (const float *)in, (short *)out, channels, n,
                           scale, t);
    LLZ_LAUNCH_CHECK("k_pcm_transpose");
    return LLZ_OK;
}

} // namespace

extern "C" int llzs_pcm_deinterleave_i16_f32(const short *in, float *out, int channels, long n, float scale,
                                             void *stream)
{
    return pcm_launch(true, in, out, channels, n, scale, stream);
}

extern "C" int llzs_pcm_interleave_f32_i16(const float *in, short *out, int channels, long n, float scale, void *stream)
{
    return pcm_launch(false, in, out, channels, n, scale, stream);
}
