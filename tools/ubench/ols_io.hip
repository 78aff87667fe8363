// micro-benchmark: the overlap-save kernel's memory skeleton (a half-wave walks 16-job segments of a channel, the next
// job's samples requested one job ahead, every sample read once and written once) under different access shapes:
//   W=1  one dword per lane, 128 B contiguous per half-wave instruction (the shipped kernel's shape)
//   W=2  8 B per lane, 256 B per half-wave instruction
//   W=4  16 B per lane, 512 B per half-wave instruction (lane l: bytes 16 l of the run)
//   W=5  16 B per lane in the quad-transposed order (lane 4q+i: row i of four 128-B rows, 16-B column q)
// with and without a stand-in for the transforms' vector work (WORK sweeps of one FMA per held dword).
// usage: ols_io [channels] -> one line per variant: ms, GB/s (8 B per sample)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int JOB = 1536, SEG = 16;

template <int W>
__device__ __forceinline__ void ld(float (&r)[48], const float *p, int l5)
{
    if (W == 1) {
#pragma unroll
        for (int i = 0; i < 48; i++) r[i] = __builtin_nontemporal_load(p + 32 * i + l5);
    } else if (W == 2) {
#pragma unroll
        for (int i = 0; i < 24; i++) {
            const f2 v = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(p + 64 * i + 2 * l5));
            r[2 * i] = v.x; r[2 * i + 1] = v.y;
        }
    } else {
        const int off = W == 4 ? 4 * l5 : 32 * (l5 & 3) + 4 * (l5 >> 2);
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p + 128 * i + off));
            r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
        }
    }
}

template <int W>
__device__ __forceinline__ void st(const float (&r)[48], float *p, int l5)
{
    if (W == 1) {
#pragma unroll
        for (int i = 0; i < 48; i++) __builtin_nontemporal_store(r[i], p + 32 * i + l5);
    } else if (W == 2) {
#pragma unroll
        for (int i = 0; i < 24; i++) {
            f2 v; v.x = r[2 * i]; v.y = r[2 * i + 1];
            __builtin_nontemporal_store(v, reinterpret_cast<f2 *>(p + 64 * i + 2 * l5));
        }
    } else {
        const int off = W == 4 ? 4 * l5 : 32 * (l5 & 3) + 4 * (l5 >> 2);
#pragma unroll
        for (int i = 0; i < 12; i++) {
            f4 v; v.x = r[4 * i]; v.y = r[4 * i + 1]; v.z = r[4 * i + 2]; v.w = r[4 * i + 3];
            __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p + 128 * i + off));
        }
    }
}

template <int W, int WORK>
__global__ void __launch_bounds__(256) k_walk(const float *__restrict__ in, float *__restrict__ out, long nsegs, float c)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, l5 = lane & 31;
    const long halves = (long)gridDim.x * 8;
    const long first = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half;
    if (first >= nsegs) return;
    float nxt[48];
    ld<W>(nxt, in + first * SEG * JOB, l5);
    for (long seg = first; seg < nsegs; seg += halves) {
        const long nseg = seg + halves < nsegs ? seg + halves : seg;      // tail: re-read own segment start (harmless)
#pragma unroll 1
        for (int j = 0; j < SEG; j++) {
            float cur[48];
#pragma unroll
            for (int i = 0; i < 48; i++) cur[i] = nxt[i];
            const float *np = j + 1 < SEG ? in + (seg * SEG + j + 1) * JOB : in + nseg * SEG * JOB;
            ld<W>(nxt, np, l5);
#pragma unroll 1
            for (int w = 0; w < WORK; w++) {
#pragma unroll
                for (int i = 0; i < 48; i++) cur[i] = __builtin_fmaf(cur[i], c, cur[(i + 1) % 48]);
            }
            st<W>(cur, out + (seg * SEG + j) * JOB, l5);
        }
    }
}

template <int W, int WORK>
void run(const char *name, const float *in, float *out, long nsegs, int per_cu)
{
    const int blocks = 256 * per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_walk<W, WORK>), dim3(blocks), dim3(256), 0, 0, in, out, nsegs, 0.5f);
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k_walk<W, WORK>), dim3(blocks), dim3(256), 0, 0, in, out, nsegs, 0.5f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-10s work=%2d wg/cu=%d: %.3f ms  %.0f GB/s\n", name, WORK, per_cu, ms, 8.0 * nsegs * SEG * JOB / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const long channels = argc > 1 ? atol(argv[1]) : 4096;
    const long nsegs = channels * ((1 << 20) / (SEG * JOB));
    const size_t bytes = (size_t)nsegs * SEG * JOB * 4;
    float *in, *out;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 0, bytes);
    for (int per_cu : {2, 3, 4}) {
        run<1, 0>("dword", in, out, nsegs, per_cu);
        run<2, 0>("dwordx2", in, out, nsegs, per_cu);
        run<4, 0>("dwordx4", in, out, nsegs, per_cu);
        run<5, 0>("dwordx4q", in, out, nsegs, per_cu);
    }
    for (int per_cu : {2, 3}) {
        run<1, 40>("dword", in, out, nsegs, per_cu);
        run<4, 40>("dwordx4", in, out, nsegs, per_cu);
        run<5, 40>("dwordx4q", in, out, nsegs, per_cu);
    }
    run<1, 24>("dword", in, out, nsegs, 2);
    run<5, 24>("dwordx4q", in, out, nsegs, 2);
    hipFree(in); hipFree(out);
    return 0;
}
