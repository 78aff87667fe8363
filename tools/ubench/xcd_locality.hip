// micro-benchmark: does the non-persistent 4 KB-tile copy (6.5 TB/s, the fastest shape measured on these boxes) owe its rate to
// which XCD touches which 4 KB granule?  Workgroups are dealt to the 8 XCDs round-robin (workgroup b -> XCD b % 8); variants of
// a 16 GiB -> 16 GiB copy, 4 KB (256 x 16 B) per workgroup step:
//   rot r     workgroup b copies tile (b & ~7) | ((b + r) & 7): the same tiles, another XCD for each
//   eighths   workgroup b copies tile (b % 8) * (n / 8) + b / 8: every XCD streams its own contiguous eighth of the buffers
//   own U     workgroup b copies U tiles that are all = b % 8 (mod 8): (b % 8) + 8 (U (b / 8) + i)   (32 KB stride)
//   run U     workgroup b copies U consecutive tiles (stream_shapes.hip's "tile U")
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int U>
__global__ void __launch_bounds__(256) k(const f4 *__restrict__ in, f4 *__restrict__ out, long ntiles, int r)
{
    const long b = blockIdx.x;
    f4 v[U];
    long t[U];
#pragma unroll
    for (int i = 0; i < U; i++) {
        if (MODE == 0) t[i] = (b & ~7L) | ((b + r) & 7);
        else if (MODE == 1) t[i] = (b % 8) * (ntiles / 8) + b / 8;
        else if (MODE == 2) t[i] = (b % 8) + 8 * (U * (b / 8) + i);
        else t[i] = b * U + i;
    }
#pragma unroll
    for (int i = 0; i < U; i++) v[i] = __builtin_nontemporal_load(in + t[i] * 256 + threadIdx.x);
#pragma unroll
    for (int i = 0; i < U; i++) __builtin_nontemporal_store(v[i], out + t[i] * 256 + threadIdx.x);
}

template <typename F>
static void timeit(const char *name, long bytes, F launch)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch();
    (void)hipEventRecord(a);
    for (int i = 0; i < 5; i++) launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-22s %.3f ms  %.0f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6);
    fflush(stdout);
}

int main()
{
    const long ntiles = 4L << 20;                      // 4 Mi tiles of 4 KB = 16 GiB
    const long bytes = ntiles * 4096;
    f4 *in, *out;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(in, 1, bytes);
    (void)hipDeviceSynchronize();
    char name[64];
    for (int r = 0; r < 8; r++) {
        snprintf(name, sizeof name, "rot %d", r);
        timeit(name, bytes, [&] { hipLaunchKernelGGL((k<0, 1>), dim3(ntiles), dim3(256), 0, 0, in, out, ntiles, r); });
    }
    timeit("eighths", bytes, [&] { hipLaunchKernelGGL((k<1, 1>), dim3(ntiles), dim3(256), 0, 0, in, out, ntiles, 0); });
    timeit("own 2", bytes, [&] { hipLaunchKernelGGL((k<2, 2>), dim3(ntiles / 2), dim3(256), 0, 0, in, out, ntiles, 0); });
    timeit("own 4", bytes, [&] { hipLaunchKernelGGL((k<2, 4>), dim3(ntiles / 4), dim3(256), 0, 0, in, out, ntiles, 0); });
    timeit("own 8", bytes, [&] { hipLaunchKernelGGL((k<2, 8>), dim3(ntiles / 8), dim3(256), 0, 0, in, out, ntiles, 0); });
    timeit("run 2", bytes, [&] { hipLaunchKernelGGL((k<3, 2>), dim3(ntiles / 2), dim3(256), 0, 0, in, out, ntiles, 0); });
    timeit("run 4", bytes, [&] { hipLaunchKernelGGL((k<3, 4>), dim3(ntiles / 4), dim3(256), 0, 0, in, out, ntiles, 0); });
    timeit("run 8", bytes, [&] { hipLaunchKernelGGL((k<3, 8>), dim3(ntiles / 8), dim3(256), 0, 0, in, out, ntiles, 0); });
    timeit("rot 0 again", bytes, [&] { hipLaunchKernelGGL((k<0, 1>), dim3(ntiles), dim3(256), 0, 0, in, out, ntiles, 0); });
    (void)hipFree(in); (void)hipFree(out);
    return 0;
}
