// pcm.hip -- interleaved int16 <-> planar float32.  Pure data movement: 2 B + 4 B per
// sample-channel, bound by HBM.  A workgroup transposes a tile of TC channels x TS samples (TC*TS <= 4096) through LDS
// so that both sides see contiguous runs: TC int16 per sample on the interleaved side (whole rows when channels <= 64),
// TS floats per channel on the planar side.
#include "common.hpp"

namespace {

constexpr int PCM_THREADS = 256;
constexpr int PCM_TILE = 4096;
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
    // the interleaved side moves 4 int16 (8 bytes) per lane when the tile's channel run allows it (tc % 4 == 0 and
    // the rows are 8-byte aligned): a wave then covers 512 contiguous bytes instead of 128
    const bool vec4 = (t.tc & 3) == 0 && (channels & 3) == 0 && tc == t.tc &&
                      ((reinterpret_cast<uintptr_t>(DEINTERLEAVE ? (const void *)ileaved_in : (const void *)ileaved_out) & 7) == 0);
    // whole-row tiles with a channel count that is not a multiple of 4 (stereo, 5.1 ...): the tile's interleaved bytes
    // are one contiguous run, read 8 bytes at a time and split per element
    const bool flat4 = !vec4 && t.tc == channels && (total & 3) == 0 && ((i0 * channels) & 3) == 0 &&
                       ((reinterpret_cast<uintptr_t>(DEINTERLEAVE ? (const void *)ileaved_in : (const void *)ileaved_out) & 7) == 0);
    // the planar side moves 4 floats per lane when the tile is full along time and the rows are 16-byte aligned
    const bool pvec4 = (t.ts & 3) == 0 && (n & 3) == 0 && ts == t.ts &&
                       ((reinterpret_cast<uintptr_t>(DEINTERLEAVE ? (const void *)planar_out : (const void *)planar_in) & 15) == 0);
    if (DEINTERLEAVE) {
        if (vec4) {
            for (int e4 = tid; e4 < total / 4; e4 += PCM_THREADS) {    // e = s*TC + c, 4 consecutive channels
                const int e = e4 * 4;
                const int s = div_magic(e, t.tc_magic, t.tc), c = e - s * t.tc;
                if (s < ts) {
                    const short4 v = *reinterpret_cast<const short4 *>(&ileaved_in[(i0 + s) * channels + c0 + c]);
                    const int a = c * t.ts + s;
                    tile[a + (a >> 5)] = (float)v.x * scale;
                    tile[(a + t.ts) + ((a + t.ts) >> 5)] = (float)v.y * scale;
                    tile[(a + 2 * t.ts) + ((a + 2 * t.ts) >> 5)] = (float)v.z * scale;
                    tile[(a + 3 * t.ts) + ((a + 3 * t.ts) >> 5)] = (float)v.w * scale;
                }
            }
        } else if (flat4) {
            const int live = ts * t.tc;                                // the tile's rows are one contiguous run
            const short *src = ileaved_in + i0 * channels;
            for (int e4 = tid; e4 < total / 4; e4 += PCM_THREADS) {
                const int e = e4 * 4;
                if (e + 3 < live) {
                    const short4 v = *reinterpret_cast<const short4 *>(&src[e]);
                    const short q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int s = div_magic(e + j, t.tc_magic, t.tc), c = e + j - s * t.tc;
                        const int a = c * t.ts + s;
                        tile[a + (a >> 5)] = (float)q[j] * scale;
                    }
                } else {
                    for (int j = 0; j < 4 && e + j < live; j++) {
                        const int s = div_magic(e + j, t.tc_magic, t.tc), c = e + j - s * t.tc;
                        const int a = c * t.ts + s;
                        tile[a + (a >> 5)] = (float)src[e + j] * scale;
                    }
                }
            }
        } else {
            for (int e = tid; e < total; e += PCM_THREADS) {           // e = s*TC + c: contiguous along channels
                const int s = div_magic(e, t.tc_magic, t.tc), c = e - s * t.tc;
                if (s < ts && c < tc) {
                    const int a = c * t.ts + s;
                    tile[a + (a >> 5)] = (float)ileaved_in[(i0 + s) * channels + c0 + c] * scale;
                }
            }
        }
        __syncthreads();
        if (pvec4) {
            for (int e4 = tid; e4 < total / 4; e4 += PCM_THREADS) {    // e = c*TS + s, 4 consecutive samples
                const int e = e4 * 4;
                const int c = div_magic(e, t.ts_magic, t.ts), s = e - c * t.ts;
                if (c < tc) {
                    const int a = e + (e >> 5);
                    const f32x4 v = {tile[a], tile[a + 1], tile[a + 2], tile[a + 3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(&planar_out[(size_t)(c0 + c) * n + i0 + s]));
                }
            }
        } else {
            for (int e = tid; e < total; e += PCM_THREADS) {           // e = c*TS + s: contiguous along time
                const int c = div_magic(e, t.ts_magic, t.ts), s = e - c * t.ts;
                if (s < ts && c < tc) planar_out[(size_t)(c0 + c) * n + i0 + s] = tile[e + (e >> 5)];
            }
        }
    } else {
        if (pvec4) {
            for (int e4 = tid; e4 < total / 4; e4 += PCM_THREADS) {
                const int e = e4 * 4;
                const int c = div_magic(e, t.ts_magic, t.ts), s = e - c * t.ts;
                if (c < tc) {
                    const f32x4 v = __builtin_nontemporal_load(
                        reinterpret_cast<const f32x4 *>(&planar_in[(size_t)(c0 + c) * n + i0 + s]));
                    const int a = e + (e >> 5);
                    tile[a] = v.x; tile[a + 1] = v.y; tile[a + 2] = v.z; tile[a + 3] = v.w;
                }
            }
        } else {
            for (int e = tid; e < total; e += PCM_THREADS) {
                const int c = div_magic(e, t.ts_magic, t.ts), s = e - c * t.ts;
                if (s < ts && c < tc) tile[e + (e >> 5)] = planar_in[(size_t)(c0 + c) * n + i0 + s];
            }
        }
        __syncthreads();
        auto quant = [&](int a) {
            float y = tile[a + (a >> 5)] * scale;
            y = fminf(fmaxf(y, -32768.f), 32767.f);                    // llz_resample.c:596-599
            return (short)(int)y;                                      // :601, truncation toward zero
        };
        if (vec4) {
            for (int e4 = tid; e4 < total / 4; e4 += PCM_THREADS) {
                const int e = e4 * 4;
                const int s = div_magic(e, t.tc_magic, t.tc), c = e - s * t.tc;
                if (s < ts) {
                    const int a = c * t.ts + s;
                    short4 v;
                    v.x = quant(a); v.y = quant(a + t.ts); v.z = quant(a + 2 * t.ts); v.w = quant(a + 3 * t.ts);
                    *reinterpret_cast<short4 *>(&ileaved_out[(i0 + s) * channels + c0 + c]) = v;
                }
            }
        } else if (flat4) {
            const int live = ts * t.tc;
            short *dst = ileaved_out + i0 * channels;
            for (int e4 = tid; e4 < total / 4; e4 += PCM_THREADS) {
                const int e = e4 * 4;
                short q[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int s = div_magic(e + j, t.tc_magic, t.tc), c = e + j - s * t.tc;
                    q[j] = quant(c * t.ts + s);
                }
                if (e + 3 < live) {
                    short4 v; v.x = q[0]; v.y = q[1]; v.z = q[2]; v.w = q[3];
                    *reinterpret_cast<short4 *>(&dst[e]) = v;
                } else {
                    for (int j = 0; j < 4 && e + j < live; j++) dst[e + j] = q[j];
                }
            }
        } else {
            for (int e = tid; e < total; e += PCM_THREADS) {
                const int s = div_magic(e, t.tc_magic, t.tc), c = e - s * t.tc;
                if (s < ts && c < tc) ileaved_out[(i0 + s) * channels + c0 + c] = quant(c * t.ts + s);
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
