// micro-benchmark (follow-up of walk_substreams.hip): the persistent walk copy where every request PAIR of a half-wave goes to two
// places exactly D bytes apart (the overlap-save kernel could do this: the two real blocks packed into one complex transform can
// come from anywhere).  The buffer is units of 2 D bytes; a work item is a region of R bytes in the first half of a unit
// and the region D bytes further on; a job moves 4 KB of each (32 rows of 128 B, requests alternating between the two).
//   D = 0: the plain walk (a job = 8 KB contiguous, items = 2 R contiguous)
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void __launch_bounds__(256) k_walk(const float *__restrict__ in, float *__restrict__ out, long nitems, long Dfl, long Rfl,
                                              long per_unit, int steps)
{
    const int lane = threadIdx.x & 63, half = lane >> 5, l5 = lane & 31;
    const long halves = (long)gridDim.x * 8;
    const long first = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half;
    if (first >= nitems) return;
    float nxt[64];
    auto base = [&](long it) { return Dfl ? (it / per_unit) * 2 * Dfl + (it % per_unit) * Rfl : it * 2 * Rfl; };
    auto ld = [&](long it, int step) {
        const float *p = in + base(it) + (long)step * (Dfl ? 1024 : 2048) + l5;
#pragma unroll
        for (int i = 0; i < 64; i++) nxt[i] = __builtin_nontemporal_load(Dfl ? p + (i & 1) * Dfl + (i >> 1) * 32 : p + i * 32);
    };
    ld(first, 0);
    for (long it = first; it < nitems; it += halves) {
        const long nit = it + halves < nitems ? it + halves : it;
#pragma unroll 1
        for (int s = 0; s < steps; s++) {
            float cur[64];
#pragma unroll
            for (int i = 0; i < 64; i++) cur[i] = nxt[i];
            if (s + 1 < steps) ld(it, s + 1); else ld(nit, 0);
            float *q = out + base(it) + (long)s * (Dfl ? 1024 : 2048) + l5;
#pragma unroll
            for (int i = 0; i < 64; i++) __builtin_nontemporal_store(cur[i], Dfl ? q + (i & 1) * Dfl + (i >> 1) * 32 : q + i * 32);
        }
    }
}

static void run(const float *in, float *out, long bytes, long D, long R)
{
    // every access must stay inside the buffers: whole units of 2 D bytes only (a D that does not divide the buffer leaves a
    // tail untouched; the first version of this file ran over the end there and faulted)
    const long per_unit = D ? D / R : 1;
    const long nitems = D ? (bytes / (2 * D)) * per_unit : bytes / (2 * R);
    const int steps = (int)(R / 4096);
    if ((D && (D % R || R % 4096)) || nitems < 1) { printf("D=%ld R=%ld: skipped (shape)\n", D, R); return; }
    {   // host-side bounds check of the LAST item's LAST request before anything is launched (floats): item base + last step +
        // the farthest row of a request group (+ the partner place D further on) + one 32-float row
        const long Dfl = D / 4, Rfl = R / 4, last = nitems - 1;
        const long base = D ? (last / per_unit) * 2 * Dfl + (last % per_unit) * Rfl : last * 2 * Rfl;
        const long reach = D ? base + (long)(steps - 1) * 1024 + Dfl + 31 * 32 + 32 : base + (long)(steps - 1) * 2048 + 63 * 32 + 32;
        if (reach > bytes / 4) {
            printf("D=%ld R=%ld: REFUSED, last access would end at float %ld of %ld\n", D, R, reach, bytes / 4);
            return;
        }
    }
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto go = [&] { hipLaunchKernelGGL(k_walk, dim3(512), dim3(256), 0, 0, in, out, nitems, D / 4, R / 4, per_unit, steps); };
    go();
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; r++) go();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double moved = D ? (double)nitems * 2 * R : (double)bytes;
    printf("D=%6ld KB R=%4ld KB: %.3f ms  %.0f GB/s\n", D >> 10, R >> 10, ms, 2.0 * moved / ms / 1e6);
    fflush(stdout);
}

int main()
{
    const long bytes = 16L << 30;
    float *in, *out;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(in, 0, bytes);
    run(in, out, bytes, 0, 16L << 10);
    run(in, out, bytes, 0, 48L << 10);
    for (long kb : {16L, 32L, 64L, 128L, 256L, 512L, 1024L, 2048L, 4096L, 8192L, 16384L}) run(in, out, bytes, kb << 10, 16L << 10);
    run(in, out, bytes, 48L << 10, 16L << 10);      // 2 D does not divide the buffer: whole units only, the tail stays untouched
    run(in, out, bytes, 32L << 10, 32L << 10);
    run(in, out, bytes, 4096L << 10, 64L << 10);
    run(in, out, bytes, 0, 16L << 10);
    (void)hipFree(in); (void)hipFree(out);
    return 0;
}
