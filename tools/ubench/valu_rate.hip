// micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_pk_add_f32 vs v_fma_f64 on gfx950 (register-only loops)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const float c = 1.0001f, e = 0.0001f;
    const f2 c2 = {c, c}, e2 = {e, e};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "v"(e));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(c), "v"(e));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(c), "v"(e));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(c), "v"(e));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(c), "v"(e));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(c), "v"(e));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(c), "v"(e));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(c), "v"(e));
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p4) : "v"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p5) : "v"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p6) : "v"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p7) : "v"(c2), "v"(e2));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(e2));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p1) : "v"(e2));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2) : "v"(e2));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p3) : "v"(e2));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p4) : "v"(e2));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p5) : "v"(e2));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p6) : "v"(e2));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p7) : "v"(e2));
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(e));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(e));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(e));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a3) : "v"(e));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a4) : "v"(e));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a5) : "v"(e));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a6) : "v"(e));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a7) : "v"(e));
            }
        } else if (MODE == 6) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(c), "v"(e));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"(c), "v"(e));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a2) : "v"(c), "v"(e));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a3) : "v"(c), "v"(e));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a4) : "v"(c), "v"(e));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a5) : "v"(c), "v"(e));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a6) : "v"(c), "v"(e));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a7) : "v"(c), "v"(e));
            }
        } else if (MODE == 7) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_fmamk_f32 %0, %0, 0x3f800347, %1" : "+v"(a0) : "v"(e));
                asm volatile("v_fmamk_f32 %0, %0, 0x3f800347, %1" : "+v"(a1) : "v"(e));
                asm volatile("v_fmamk_f32 %0, %0, 0x3f800347, %1" : "+v"(a2) : "v"(e));
                asm volatile("v_fmamk_f32 %0, %0, 0x3f800347, %1" : "+v"(a3) : "v"(e));
                asm volatile("v_fmamk_f32 %0, %0, 0x3f800347, %1" : "+v"(a4) : "v"(e));
                asm volatile("v_fmamk_f32 %0, %0, 0x3f800347, %1" : "+v"(a5) : "v"(e));
                asm volatile("v_fmamk_f32 %0, %0, 0x3f800347, %1" : "+v"(a6) : "v"(e));
                asm volatile("v_fmamk_f32 %0, %0, 0x3f800347, %1" : "+v"(a7) : "v"(e));
            }
        } else if (MODE == 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "s"(c), "v"(e));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "s"(c), "v"(e));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "s"(c), "v"(e));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "s"(c), "v"(e));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "s"(c), "v"(e));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "s"(c), "v"(e));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "s"(c), "v"(e));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "s"(c), "v"(e));
            }
        } else if (MODE == 9) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "s"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "s"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "s"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "s"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p4) : "s"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p5) : "s"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p6) : "s"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p7) : "s"(c2), "v"(e2));
            }
        } else if (MODE == 10) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(c));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a1) : "v"(c));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(c));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a3) : "v"(c));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a4) : "v"(c));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a5) : "v"(c));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a6) : "v"(c));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a7) : "v"(c));
            }
        } else if (MODE == 11) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(c2));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p1) : "v"(c2));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p2) : "v"(c2));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p3) : "v"(c2));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p4) : "v"(c2));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p5) : "v"(c2));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p6) : "v"(c2));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p7) : "v"(c2));
            }
        } else if (MODE == 12) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a0) : "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a1) : "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a2) : "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a3) : "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a4) : "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a5) : "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a6) : "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a7) : "v"(c));
            }
        } else if (MODE == 13) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "s"(c), "v"(e));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "s"(c), "v"(e));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a2) : "s"(c), "v"(e));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a3) : "s"(c), "v"(e));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a4) : "s"(c), "v"(e));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a5) : "s"(c), "v"(e));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a6) : "s"(c), "v"(e));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a7) : "s"(c), "v"(e));
            }
        } else if (MODE == 14) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p0) : "s"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p1) : "s"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p2) : "s"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p3) : "s"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p4) : "s"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p5) : "s"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p6) : "s"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p7) : "s"(c2), "v"(e2));
            }
        } else if (MODE == 15) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p0) : "v"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p1) : "v"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p2) : "v"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p3) : "v"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p4) : "v"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p5) : "v"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p6) : "v"(c2), "v"(e2));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p7) : "v"(c2), "v"(e2));
            }
        } else if (MODE == 5) {
            // complex multiply p *= c2 on the packed pipe: t = (p.x c.x, p.x c.y); p = (-p.y c.y + t.x, p.y c.x + t.y)
#pragma unroll
            for (int u = 0; u < 8; u++) {
#define CM(P) { f2 t; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(P), "v"(c2)); \
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(P) : "v"(P), "v"(c2), "v"(t)); }
                CM(p0) CM(p1) CM(p2) CM(p3)
            }
        } else {
            const double dc = 1.0001, de = 0.0001;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(dc), "v"(de));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d1) : "v"(dc), "v"(de));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d2) : "v"(dc), "v"(de));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d3) : "v"(dc), "v"(de));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d4) : "v"(dc), "v"(de));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d5) : "v"(dc), "v"(de));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d6) : "v"(dc), "v"(de));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d7) : "v"(dc), "v"(de));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x +
                                          p6.x + p7.x + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

template <int MODE>
void run(const char *name, int waves_per_simd, float *d)
{
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;          // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_instr_per_simd = (double)iters * 64 * waves_per_simd;
    printf("%-14s %d wave/SIMD: %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name,
           waves_per_simd, ms, ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
}

// semantics of the op_sel / neg_lo forms used for complex arithmetic on the packed pipe
__global__ void k_sem(float *o)
{
    f2 x = {3.f, 5.f}, w = {7.f, 11.f}, t, r, s, d;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(x), "v"(w));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(x), "v"(w), "v"(t));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(s) : "v"(x), "v"(w));
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(w));
    o[0] = t.x; o[1] = t.y; o[2] = r.x; o[3] = r.y; o[4] = s.x; o[5] = s.y; o[6] = d.x; o[7] = d.y;
}

int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    {
        float h[8];
        hipLaunchKernelGGL(k_sem, dim3(1), dim3(1), 0, 0, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("x=(3,5) w=(7,11): t=(%g,%g) [21,33]  x*w=(%g,%g) [-34,68]  x+w=(%g,%g) [10,16]  x-w=(%g,%g) [-4,-6]\n", h[0], h[1],
               h[2], h[3], h[4], h[5], h[6], h[7]);
    }
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w, d);
        run<1>("v_pk_fma_f32", w, d);
        run<2>("v_pk_add_f32", w, d);
        run<3>("v_add_f32", w, d);
        run<4>("v_fma_f64", w, d);
        run<5>("cmul(pk x2)/2", w, d);
        run<6>("v_fmac_f32", w, d);
        run<7>("v_fmamk_f32", w, d);
        run<8>("v_fma_f32 sgpr", w, d);
        run<9>("v_pk_fma sgpr", w, d);
        run<10>("v_mul_f32", w, d);
        run<11>("v_pk_mul_f32", w, d);
        run<12>("v_fma_f32 2rd", w, d);
        run<13>("v_fmac s,v", w, d);
        run<14>("v_pk_fma s.xx,v", w, d);
        run<15>("v_pk_fma v.xx,v", w, d);
    }
    return 0;
}
