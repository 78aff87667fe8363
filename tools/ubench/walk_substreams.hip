// micro-benchmark: the persistent walk copy (a half-wave streams its own region, next job requested one job ahead, one dword per
// lane) where the 64 rows (128 B each) of a job are dealt round-robin to K sub-streams D bytes apart instead of being 8 KB
// contiguous.  wave_stride.hip shows that a wave's back-to-back requests want to be 32-64 KB apart (HBM stacks look interleaved
// at 32 KB: eight requests into one 8 KB run queue on one stack); this asks whether the same holds for the overlap-save
// kernel's shape.  A half-wave owns a super-block of K D bytes at a time: sub-stream k = its k-th slice of D bytes.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int K>
__global__ void __launch_bounds__(256) k_walk(const float *__restrict__ in, float *__restrict__ out, long nblocks, long Dfl,
                                              long blockfl, int steps)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, l5 = lane & 31;
    const long halves = (long)gridDim.x * 8;
    const long first = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half;
    if (first >= nblocks) return;
    constexpr int RPS = 64 / K;                          // rows per sub-stream and job
    float nxt[64];
    auto ld = [&](long blk, int step) {
        const float *p = in + blk * blockfl + (long)step * RPS * 32 + l5;
#pragma unroll
        for (int i = 0; i < 64; i++) nxt[i] = __builtin_nontemporal_load(p + (i % K) * Dfl + (i / K) * 32);
    };
    ld(first, 0);
    for (long blk = first; blk < nblocks; blk += halves) {
        const long nblk = blk + halves < nblocks ? blk + halves : blk;
#pragma unroll 1
        for (int s = 0; s < steps; s++) {
            float cur[64];
#pragma unroll
            for (int i = 0; i < 64; i++) cur[i] = nxt[i];
            if (s + 1 < steps) ld(blk, s + 1); else ld(nblk, 0);
            float *q = out + blk * blockfl + (long)s * RPS * 32 + l5;
#pragma unroll
            for (int i = 0; i < 64; i++) __builtin_nontemporal_store(cur[i], q + (i % K) * Dfl + (i / K) * 32);
        }
    }
}

template <int K>
static void run(const float *in, float *out, long bytes, long D)
{
    const long blockbytes = K == 1 ? 256 * 1024 : K * D;        // K = 1: the same 256 KB regions, walked contiguously
    const long nblocks = bytes / blockbytes;
    const int steps = (int)(blockbytes / 8192);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto go = [&] { hipLaunchKernelGGL(k_walk<K>, dim3(512), dim3(256), 0, 0, in, out, nblocks, D / 4, blockbytes / 4, steps); };
    go();
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; r++) go();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("K=%d D=%4ld KB: %.3f ms  %.0f GB/s\n", K, D >> 10, ms, 2.0 * bytes / ms / 1e6);
    fflush(stdout);
}

int main()
{
    const long bytes = 16L << 30;
    float *in, *out;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(in, 0, bytes);
    run<1>(in, out, bytes, 0);
    for (long D : {32L << 10, 64L << 10}) { run<2>(in, out, bytes, D); run<4>(in, out, bytes, D); run<8>(in, out, bytes, D); }
    run<8>(in, out, bytes, 16L << 10);
    run<2>(in, out, bytes, 128L << 10);
    run<1>(in, out, bytes, 0);
    for (long kb : {16, 20, 24, 28, 32, 36, 40, 44, 48, 56, 96}) run<2>(in, out, bytes, kb << 10);
    run<2>(in, out, bytes, 32L << 10);
    (void)hipFree(in); (void)hipFree(out);
    return 0;
}
