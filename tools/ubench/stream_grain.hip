// micro-benchmark: why does a non-persistent 4 KB-per-workgroup copy reach 6.4 TB/s on a box where the overlap-save kernel's
// walk shape (a half-wave requests 6 KB, then stores 6 KB) stops at 5.6-5.7?  Variants of a 16 GiB -> 16 GiB copy:
//   tile U      non-persistent, U x 16 B per lane, all loads then all stores                       (stream_shapes.hip)
//   tileseq U   the same bytes per workgroup, but load -> wait -> store one 16-B piece at a time (1 KB in flight per wave)
//   gs W        persistent grid-stride, 1 piece per iteration, W workgroups per CU
//   walk P D    persistent walk of 96 KB segments per half-wave, dword accesses, the next job's 48 loads issued in P groups
//               with vector work between them (total WORK sweeps per job); D = 1: the job's 48 stores spread the same way
//               over the following job's groups (needs the results to stay in registers: what it would buy)
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

template <int U>
__global__ void __launch_bounds__(256) k_tileseq(const f4 *__restrict__ in, f4 *__restrict__ out)
{
    const long base = (long)blockIdx.x * (256 * U) + threadIdx.x;
#pragma unroll 1
    for (int i = 0; i < U; i++) {
        f4 v = __builtin_nontemporal_load(in + base + 256 * i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_nontemporal_store(v, out + base + 256 * i);
    }
}

__global__ void __launch_bounds__(256) k_gs(const f4 *__restrict__ in, f4 *__restrict__ out, long n4)
{
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        f4 v = __builtin_nontemporal_load(in + i);
        __builtin_nontemporal_store(v, out + i);
    }
}

constexpr int JOB = 1536, SEG = 16;

template <int P, int SPREAD_ST, int WORK>
__global__ void __launch_bounds__(256) k_walk(const float *__restrict__ in, float *__restrict__ out, long nsegs, float c)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, l5 = lane & 31;
    const long halves = (long)gridDim.x * 8;
    const long first = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half;
    if (first >= nsegs) return;
    constexpr int G = 48 / P;
    float nxt[48], res[48];
#pragma unroll
    for (int i = 0; i < 48; i++) { nxt[i] = __builtin_nontemporal_load(in + first * SEG * JOB + 32 * i + l5); res[i] = 0.f; }
    float *prev_out = nullptr;
    for (long seg = first; seg < nsegs; seg += halves) {
        const long nseg = seg + halves < nsegs ? seg + halves : seg;
#pragma unroll 1
        for (int j = 0; j < SEG; j++) {
            float cur[48];
#pragma unroll
            for (int i = 0; i < 48; i++) cur[i] = nxt[i];
            const float *np = j + 1 < SEG ? in + (seg * SEG + j + 1) * JOB : in + nseg * SEG * JOB;
            float *op = out + (seg * SEG + j) * JOB;
#pragma unroll
            for (int p = 0; p < P; p++) {
#pragma unroll
                for (int i = 0; i < G; i++) nxt[p * G + i] = __builtin_nontemporal_load(np + 32 * (p * G + i) + l5);
                if (SPREAD_ST && prev_out) {
#pragma unroll
                    for (int i = 0; i < G; i++) __builtin_nontemporal_store(res[p * G + i], prev_out + 32 * (p * G + i) + l5);
                }
#pragma unroll 1
                for (int w = 0; w < WORK / P; w++) {
#pragma unroll
                    for (int i = 0; i < 48; i++) cur[i] = __builtin_fmaf(cur[i], c, cur[(i + 1) % 48]);
                }
            }
            if (SPREAD_ST) {
#pragma unroll
                for (int i = 0; i < 48; i++) res[i] = cur[i];
                prev_out = op;
            } else {
#pragma unroll
                for (int i = 0; i < 48; i++) __builtin_nontemporal_store(cur[i], op + 32 * i + l5);
            }
        }
    }
    if (SPREAD_ST && prev_out) {
#pragma unroll
        for (int i = 0; i < 48; i++) __builtin_nontemporal_store(res[i], prev_out + 32 * i + l5);
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
    printf("%-34s %.3f ms  %.0f GB/s\n", name, ms, 8.0 * floats / ms / 1e6);
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
    timeit("tile U=1", floats, [&] { hipLaunchKernelGGL(k_tile<1>, dim3(floats / 4 / 256), dim3(256), 0, 0, i4, o4); });
    timeit("tile U=8", floats, [&] { hipLaunchKernelGGL(k_tile<8>, dim3(floats / 4 / 2048), dim3(256), 0, 0, i4, o4); });
    timeit("tileseq U=2", floats, [&] { hipLaunchKernelGGL(k_tileseq<2>, dim3(floats / 4 / 512), dim3(256), 0, 0, i4, o4); });
    timeit("tileseq U=8", floats, [&] { hipLaunchKernelGGL(k_tileseq<8>, dim3(floats / 4 / 2048), dim3(256), 0, 0, i4, o4); });
    timeit("tileseq U=32", floats, [&] { hipLaunchKernelGGL(k_tileseq<32>, dim3(floats / 4 / 8192), dim3(256), 0, 0, i4, o4); });
    for (int w : {2, 4, 8}) {
        char name[64];
        snprintf(name, sizeof name, "grid-stride persistent wg/cu=%d", w);
        timeit(name, floats, [&] { hipLaunchKernelGGL(k_gs, dim3(256 * w), dim3(256), 0, 0, i4, o4, floats / 4); });
    }
#define WALK(P, D, W) timeit("walk P=" #P " spread_st=" #D " work=" #W, floats, [&] { \
        hipLaunchKernelGGL((k_walk<P, D, W>), dim3(512), dim3(256), 0, 0, in, out, nsegs, 0.5f); })
    WALK(1, 0, 0); WALK(4, 0, 0); WALK(12, 0, 0); WALK(4, 1, 0); WALK(12, 1, 0);
    WALK(1, 0, 36); WALK(4, 0, 36); WALK(12, 0, 36); WALK(4, 1, 36); WALK(12, 1, 36);
    hipFree(in); hipFree(out);
    return 0;
}
