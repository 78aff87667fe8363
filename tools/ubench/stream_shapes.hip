// micro-benchmark: what the box's HBM delivers for a 16 GiB -> 16 GiB streaming copy under different CHIP-LEVEL shapes
// (how many concurrent sequential streams, how long the contiguous run of one wave is), all with 16 bytes per lane:
//   tile     non-persistent: workgroup b copies bytes [b T, (b+1) T), T = 256 x 16 x U (the chip works on one compact window)
//   walkhw   persistent: every HALF-wave streams its own 96 KB segments (the overlap-save kernel's shape: 4096 streams)
//   walkw    persistent: every WAVE streams its own segments (2048 streams)
//   wgstream persistent: the 4 waves of a workgroup share one stream (512 streams), job k of a round to wave k % 4
// usage: stream_shapes  -> ms and GB/s (8 B per float copied)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_fill(float *p, long n)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        unsigned u = (unsigned)i * 2654435761u;
        u ^= u >> 15; u *= 0x85EBCA6Bu; u ^= u >> 13;
        p[i] = (float)(u >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
}

template <int U>
__global__ void __launch_bounds__(256) k_tile(const f4 *__restrict__ in, f4 *__restrict__ out)
{
    const long base = (long)blockIdx.x * (256 * U) + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int i = 0; i < U; i++) v[i] = __builtin_nontemporal_load(in + base + 256 * i);
#pragma unroll
    for (int i = 0; i < U; i++) __builtin_nontemporal_store(v[i], out + base + 256 * i);
}

constexpr int JOB = 1536, SEG = 16;

// LANES lanes (32: half-wave, 64: wave, 256: workgroup) share a stream; one "job" = 1536 floats per 32 lanes
template <int LANES>
__global__ void __launch_bounds__(256) k_walk(const float *__restrict__ in, float *__restrict__ out, long nsegs)
{
    const int group = threadIdx.x / LANES, l = threadIdx.x % LANES;
    constexpr int GROUPS = 256 / LANES;
    constexpr long CHUNK = (long)JOB * (LANES / 32);                 // floats one group moves per step
    const long ngroups = (long)gridDim.x * GROUPS;
    const long first = (long)blockIdx.x * GROUPS + group;
    const long gsegs = nsegs / (LANES / 32);                         // segments of SEG steps of CHUNK floats
    if (first >= gsegs) return;
    f4 nxt[12];
    {
        const float *p = in + first * SEG * CHUNK;
#pragma unroll
        for (int i = 0; i < 12; i++) nxt[i] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p + 4 * (LANES * i + l)));
    }
    for (long seg = first; seg < gsegs; seg += ngroups) {
        const long nseg = seg + ngroups < gsegs ? seg + ngroups : seg;
#pragma unroll 1
        for (int j = 0; j < SEG; j++) {
            f4 cur[12];
#pragma unroll
            for (int i = 0; i < 12; i++) cur[i] = nxt[i];
            const float *np = j + 1 < SEG ? in + (seg * SEG + j + 1) * CHUNK : in + nseg * SEG * CHUNK;
#pragma unroll
            for (int i = 0; i < 12; i++) nxt[i] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(np + 4 * (LANES * i + l)));
            float *op = out + (seg * SEG + j) * CHUNK;
#pragma unroll
            for (int i = 0; i < 12; i++) __builtin_nontemporal_store(cur[i], reinterpret_cast<f4 *>(op + 4 * (LANES * i + l)));
        }
    }
}

template <typename F>
static void timeit(const char *name, long floats, F launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-28s %.3f ms  %.0f GB/s\n", name, ms, 8.0 * floats / ms / 1e6);
    fflush(stdout);
}

int main()
{
    const long nsegs = 4096L * 42 / 8 * 8;
    const long floats = nsegs * SEG * JOB;
    float *in, *out;
    if (hipMalloc(&in, floats * 4) != hipSuccess || hipMalloc(&out, floats * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in, floats);
    hipDeviceSynchronize();
    const f4 *i4 = (const f4 *)in;
    f4 *o4 = (f4 *)out;
    timeit("tile U=1 (4 KB/wg)", floats, [&] { hipLaunchKernelGGL(k_tile<1>, dim3(floats / 4 / 256), dim3(256), 0, 0, i4, o4); });
    timeit("tile U=2 (8 KB/wg)", floats, [&] { hipLaunchKernelGGL(k_tile<2>, dim3(floats / 4 / 512), dim3(256), 0, 0, i4, o4); });
    timeit("tile U=4 (16 KB/wg)", floats, [&] { hipLaunchKernelGGL(k_tile<4>, dim3(floats / 4 / 1024), dim3(256), 0, 0, i4, o4); });
    timeit("tile U=8 (32 KB/wg)", floats, [&] { hipLaunchKernelGGL(k_tile<8>, dim3(floats / 4 / 2048), dim3(256), 0, 0, i4, o4); });
    for (int per_cu : {1, 2, 4}) {
        char name[64];
        snprintf(name, sizeof name, "walk half-wave wg/cu=%d", per_cu);
        timeit(name, floats, [&] { hipLaunchKernelGGL(k_walk<32>, dim3(256 * per_cu), dim3(256), 0, 0, in, out, nsegs); });
        snprintf(name, sizeof name, "walk wave      wg/cu=%d", per_cu);
        timeit(name, floats, [&] { hipLaunchKernelGGL(k_walk<64>, dim3(256 * per_cu), dim3(256), 0, 0, in, out, nsegs); });
        snprintf(name, sizeof name, "walk workgroup wg/cu=%d", per_cu);
        timeit(name, floats, [&] { hipLaunchKernelGGL(k_walk<256>, dim3(256 * per_cu), dim3(256), 0, 0, in, out, nsegs); });
    }
    hipFree(in); hipFree(out);
    return 0;
}
