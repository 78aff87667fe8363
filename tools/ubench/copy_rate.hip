// micro-benchmark: streaming copy 16 GiB -> 16 GiB in the overlap-save kernel's job shape (a half-wave moves 6 KB
// contiguous per job, jobs of a channel consecutive), dword vs dwordx4 per lane, plain vs non-temporal stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int WIDTH, bool NT, int PREFETCH>
__global__ void __launch_bounds__(256) k_copy(const float *__restrict__ in, float *__restrict__ out, long njobs)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, l5 = lane & 31;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (long)gridDim.x * 4;
    // each half-wave walks a contiguous run of 16 jobs (segments dealt round robin), 1536 floats per job
    const long segs = njobs / 16;
    for (long sp = wave * 2; sp < segs; sp += waves * 2) {
        const long seg = sp + half;
        if (seg >= segs) continue;
        const float *src = in + seg * 16 * 1536;
        float *dst = out + seg * 16 * 1536;
        if (WIDTH == 1) {
            float v[48];
            for (int j = 0; j < 16; j++) {
#pragma unroll
                for (int i = 0; i < 48; i++) v[i] = src[j * 1536 + 32 * i + l5];
#pragma unroll
                for (int i = 0; i < 48; i++) {
                    if (NT) __builtin_nontemporal_store(v[i], &dst[j * 1536 + 32 * i + l5]);
                    else dst[j * 1536 + 32 * i + l5] = v[i];
                }
            }
        } else {
            f4 v[12];
            for (int j = 0; j < 16; j++) {
#pragma unroll
                for (int i = 0; i < 12; i++) v[i] = *reinterpret_cast<const f4 *>(src + j * 1536 + 4 * (32 * i + l5));
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    f4 *p = reinterpret_cast<f4 *>(dst + j * 1536 + 4 * (32 * i + l5));
                    if (NT) __builtin_nontemporal_store(v[i], p);
                    else *p = v[i];
                }
            }
        }
    }
}

// plain linear grid-stride copy, 16 B per lane, 4 loads in flight per lane
template <bool NT>
__global__ void __launch_bounds__(256) k_linear(const f4 *__restrict__ in, f4 *__restrict__ out, long n4)
{
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        if (NT) {
            __builtin_nontemporal_store(a, &out[i]); __builtin_nontemporal_store(b, &out[i + stride]);
            __builtin_nontemporal_store(c, &out[i + 2 * stride]); __builtin_nontemporal_store(d, &out[i + 3 * stride]);
        } else { out[i] = a; out[i + stride] = b; out[i + 2 * stride] = c; out[i + 3 * stride] = d; }
    }
    for (; i < n4; i += stride) out[i] = in[i];
}

template <int WIDTH, bool NT>
void run(const char *name, const float *in, float *out, long n, int blocks)
{
    const long njobs = n / 1536;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_copy<WIDTH, NT, 0>), dim3(blocks), dim3(256), 0, 0, in, out, njobs);
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k_copy<WIDTH, NT, 0>), dim3(blocks), dim3(256), 0, 0, in, out, njobs);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-28s blocks=%4d: %.3f ms  %.0f GB/s\n", name, blocks, ms, 8.0 * njobs * 1536 / ms / 1e6);
}

int main()
{
    const long n = 4096L * (1 << 20) / 1536 / 16 * 16 * 1536;
    float *in, *out;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 4);
    hipMemset(in, 1, n * 4);
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int blocks : {1024, 2048, 4096, 16384}) {
            for (int nt = 0; nt < 2; nt++) {
                if (nt) hipLaunchKernelGGL(k_linear<true>, dim3(blocks), dim3(256), 0, 0, (const f4 *)in, (f4 *)out, n / 4);
                else hipLaunchKernelGGL(k_linear<false>, dim3(blocks), dim3(256), 0, 0, (const f4 *)in, (f4 *)out, n / 4);
                hipEventRecord(a);
                for (int r = 0; r < 5; r++) {
                    if (nt) hipLaunchKernelGGL(k_linear<true>, dim3(blocks), dim3(256), 0, 0, (const f4 *)in, (f4 *)out, n / 4);
                    else hipLaunchKernelGGL(k_linear<false>, dim3(blocks), dim3(256), 0, 0, (const f4 *)in, (f4 *)out, n / 4);
                }
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
                printf("linear float4 copy %s blocks=%5d: %.3f ms  %.0f GB/s\n", nt ? "nt   " : "plain", blocks, ms, 8.0 * n / ms / 1e6);
            }
        }
        hipEventRecord(a);
        for (int r = 0; r < 5; r++) hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, 0);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("hipMemcpy DtoD: %.3f ms  %.0f GB/s\n", ms, 8.0 * n / ms / 1e6);
    }
    for (int blocks : {768}) {
        run<1, false>("dword, plain stores", in, out, n, blocks);
        run<1, true>("dword, nt stores", in, out, n, blocks);
        run<4, false>("dwordx4, plain stores", in, out, n, blocks);
        run<4, true>("dwordx4, nt stores", in, out, n, blocks);
    }
    return 0;
}
