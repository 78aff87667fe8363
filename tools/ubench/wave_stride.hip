// micro-benchmark (follow-up of xcd_locality.hip): in a non-persistent 16 GiB -> 16 GiB copy where every wave moves 8 x 1 KB,
// what matters is HOW FAR APART a wave's back-to-back requests are.  Workgroup b (4 waves) owns 32 KB; its 32 pieces of 1 KB
// are dealt to (wave w, request i) in different orders:
//   stride 1 KB   piece = 8 w + i          (each wave reads 8 KB contiguous -- the shape of the FIR / IIR kernels' bursts)
//   stride 4 KB   piece = 4 i + w          (= "run 8" of xcd_locality.hip)
//   far S         the wave's 8 requests are S bytes apart across a group of workgroups (S = 32 KB .. 1 MB), neighbours fill the gaps
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: piece = 8 w + i within the workgroup's 32 KB;  MODE 1: piece = 4 i + w;
// MODE 2: G = S / 4 KB workgroups form a group covering 8 S bytes; workgroup g of the group owns the 4 KB tile g of each of the
//         8 S-byte rows: request i of wave w = group base + i S + g 4 KB + w 1 KB
template <int MODE>
__global__ void __launch_bounds__(256) k(const f4 *__restrict__ in, f4 *__restrict__ out, long G)
{
    const long b = blockIdx.x;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    f4 v[8];
    long off[8];                                        // in f4 units
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (MODE == 0) off[i] = b * 2048 + (8 * w + i) * 64 + l;
        else if (MODE == 1) off[i] = b * 2048 + (4 * i + w) * 64 + l;
        else { const long grp = b / G, g = b % G; off[i] = grp * (8 * G * 256) + i * (G * 256) + g * 256 + w * 64 + l; }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = __builtin_nontemporal_load(in + off[i]);
#pragma unroll
    for (int i = 0; i < 8; i++) __builtin_nontemporal_store(v[i], out + off[i]);
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
    const long nwg = 512L << 10;                       // 512 Ki workgroups x 32 KB = 16 GiB
    const long bytes = nwg * 32768;
    f4 *in, *out;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(in, 1, bytes);
    (void)hipDeviceSynchronize();
    timeit("stride 1 KB", bytes, [&] { hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), 0, 0, in, out, 1L); });
    timeit("stride 4 KB", bytes, [&] { hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), 0, 0, in, out, 1L); });
    char name[64];
    for (long S : {8L << 10, 16L << 10, 32L << 10, 64L << 10, 128L << 10, 256L << 10, 1L << 20}) {
        snprintf(name, sizeof name, "far %ld KB", S >> 10);
        timeit(name, bytes, [&] { hipLaunchKernelGGL(k<2>, dim3(nwg), dim3(256), 0, 0, in, out, S / 4096); });
    }
    timeit("stride 1 KB again", bytes, [&] { hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), 0, 0, in, out, 1L); });
    (void)hipFree(in); (void)hipFree(out);
    return 0;
}
