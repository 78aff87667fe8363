// runtime.hip -- the thin runtime half of the shim: device memory, copies, streams, event timers, and the
// synthetic-PCM generator.  Everything here is plumbing for the host C layer (csrc/host/*.c).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include "common.hpp"

static thread_local char g_err[512] = "";

extern "C" void llzs_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *llz_hip_last_error(void) { return g_err; }

// ---- tuning overrides: explicit API, never the environment ---------------------------------------------------------
#include <atomic>
#include <string.h>
static std::atomic<int> g_tune[LLZS_TUNE_COUNT];
static const char *const g_tune_names[LLZS_TUNE_COUNT] = {
    "ols_wg_per_cu", "rs_generic", "rs_tiles", "rs_dec_valu", "rs_i16_path", "mfma_nacc",
    "mfma_wg_per_cu", "fft_generic", "iir_segs", "iir_unpacked", "iir_f64", "iir_pipe", "iir_wave_min_items",
    "shard_rccl", "rs_mfma_form", "ols_seg_len", "mdct_run", "mdctq_steps", "rs_i16_tiles", "rs_i16_walk", "acf_lds", "stft_full"};
namespace {
struct tune_init {
    tune_init() { for (auto &t : g_tune) t.store(-1, std::memory_order_relaxed); }
} g_tune_init;
}

extern "C" int llzs_tune(int id)
{
    return (id >= 0 && id < LLZS_TUNE_COUNT) ? g_tune[id].load(std::memory_order_relaxed) : -1;
}

extern "C" const char *llz_hip_tune_name(int index)
{
    return (index >= 0 && index < LLZS_TUNE_COUNT) ? g_tune_names[index] : nullptr;
}

extern "C" int llz_hip_tune(const char *name, int value)
{
    for (int i = 0; name && i < LLZS_TUNE_COUNT; i++)
        if (strcmp(name, g_tune_names[i]) == 0) { g_tune[i].store(value < 0 ? -1 : value); return LLZ_OK; }
    llzs_set_error("llz_hip_tune: unknown name '%s'", name ? name : "(null)");
    return LLZ_ERR_ARG;
}

extern "C" int llz_hip_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        llzs_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return LLZ_ERR_DEVICE;
    }
    return n;
}

extern "C" int llz_hip_set_device(int device)
{
    LLZ_HIP_CHECK(hipSetDevice(device));
    return LLZ_OK;
}

extern "C" int llz_hip_get_device(void)
{
    int d = -1;
    LLZ_HIP_CHECK(hipGetDevice(&d));
    return d;
}

extern "C" int llz_hip_synchronize(void *stream) { return llzs_sync(stream); }

extern "C" void *llzs_malloc(size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        llzs_set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

extern "C" void llzs_free(void *p)
{
    if (p) (void)hipFree(p);
}

extern "C" int llzs_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    if (!bytes) return LLZ_OK;
    LLZ_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    LLZ_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return LLZ_OK;
}

extern "C" int llzs_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    if (!bytes) return LLZ_OK;
    LLZ_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    LLZ_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return LLZ_OK;
}

extern "C" int llzs_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
    if (!bytes) return LLZ_OK;
    LLZ_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return LLZ_OK;
}

extern "C" int llzs_memset(void *dst, int value, size_t bytes, void *stream)
{
    if (!bytes) return LLZ_OK;
    LLZ_HIP_CHECK(hipMemsetAsync(dst, value, bytes, as_stream(stream)));
    return LLZ_OK;
}

extern "C" int llzs_sync(void *stream)
{
    LLZ_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return LLZ_OK;
}

// 1: device memory of the CURRENT device; 0: host memory (the caller's buffer gets staged); LLZ_ERR_ARG: device memory that
// lives on another GPU.  A handle binds its device before it classifies the caller's pointers, so a buffer of the wrong GPU
// is refused here with a message instead of reaching a kernel (no peer access is ever enabled by this library).
extern "C" int llzs_is_device_ptr(const void *p)
{
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();       // plain malloc'ed host memory is "invalid value" here: not an error
        return 0;
    }
    if (attr.type != hipMemoryTypeDevice) return 0;
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && attr.device != cur) {
        llzs_set_error("buffer %p lives on GPU %d, the handle (or the current device) is GPU %d", p, attr.device, cur);
        return LLZ_ERR_ARG;
    }
    return 1;
}

extern "C" void *llz_hip_malloc(size_t bytes) { return llzs_malloc(bytes); }
extern "C" void llz_hip_free(void *p) { llzs_free(p); }
extern "C" int llz_hip_upload(void *d, const void *h, size_t n) { return llzs_h2d(d, h, n, nullptr); }
extern "C" int llz_hip_download(void *h, const void *d, size_t n) { return llzs_d2h(h, d, n, nullptr); }
extern "C" int llz_hip_is_device_ptr(const void *p) { return llzs_is_device_ptr(p); }    // 1 / 0 / LLZ_ERR_ARG as above

// ---- event timer on a caller-chosen stream -------------------------------------------------------
struct llz_timer {
    hipEvent_t a, b;
};

extern "C" void *llz_hip_timer_new(void)
{
    llz_timer *t = (llz_timer *)calloc(1, sizeof(llz_timer));
    if (!t) return nullptr;
    if (hipEventCreate(&t->a) != hipSuccess || hipEventCreate(&t->b) != hipSuccess) {
        llzs_set_error("hipEventCreate failed");
        free(t);
        return nullptr;
    }
    return t;
}

extern "C" int llz_hip_timer_start(void *timer, void *stream)
{
    LLZ_HIP_CHECK(hipEventRecord(((llz_timer *)timer)->a, as_stream(stream)));
    return LLZ_OK;
}

extern "C" int llz_hip_timer_stop(void *timer, void *stream)
{
    LLZ_HIP_CHECK(hipEventRecord(((llz_timer *)timer)->b, as_stream(stream)));
    return LLZ_OK;
}

extern "C" double llz_hip_timer_ms(void *timer)
{
    llz_timer *t = (llz_timer *)timer;
    float ms = -1.f;
    if (hipEventSynchronize(t->b) != hipSuccess || hipEventElapsedTime(&ms, t->a, t->b) != hipSuccess) {
        llzs_set_error("event timing failed");
        return -1.0;
    }
    return (double)ms;
}

extern "C" void llz_hip_timer_free(void *timer)
{
    llz_timer *t = (llz_timer *)timer;
    if (!t) return;
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    free(t);
}

// ---- synthetic PCM ---------------------------------------------------------------------------------
// grid: (ceil(n / (256*4)), channels); each thread writes 4 consecutive samples (16 B / 8 B stores)
__global__ void __launch_bounds__(256)
k_synth_f32(float *__restrict__ dst, long n, long pitch, uint32_t seed, int chan0)
{
    const int c = blockIdx.y;
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    float *row = dst + (size_t)c * pitch;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t u = llz_synth_u32(seed, (uint32_t)(c + chan0), (uint32_t)(i0 + j));
        v[j] = (float)(u >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    if (i0 + 4 <= n && ((pitch & 3) == 0)) {
        *reinterpret_cast<float4 *>(row + i0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int j = 0; j < 4 && i0 + j < n; j++) row[i0 + j] = v[j];
    }
}

__global__ void __launch_bounds__(256)
k_synth_i16(short *__restrict__ dst, long n, long pitch, uint32_t seed, int chan0)
{
    const int c = blockIdx.y;
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    short *row = dst + (size_t)c * pitch;
    short v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t u = llz_synth_u32(seed, (uint32_t)(c + chan0), (uint32_t)(i0 + j));
        v[j] = (short)((int32_t)(u >> 17) - 16384);
    }
    if (i0 + 4 <= n && ((pitch & 3) == 0)) {
        *reinterpret_cast<short4 *>(row + i0) = make_short4(v[0], v[1], v[2], v[3]);
    } else {
        for (int j = 0; j < 4 && i0 + j < n; j++) row[i0 + j] = v[j];
    }
}

extern "C" int llzs_synth_f32(float *dst, int channels, long n, long pitch, unsigned seed, int chan0, void *stream)
{
    if (!dst || channels <= 0 || n <= 0 || pitch < n || channels > 65535) {
        llzs_set_error("synth_f32: bad arguments");
        return LLZ_ERR_ARG;
    }
    dim3 grid((unsigned)((n + 1023) / 1024), (unsigned)channels);
    hipLaunchKernelGGL(k_synth_f32, grid, dim3(256), 0, as_stream(stream), dst, n, pitch, seed, chan0);
    LLZ_LAUNCH_CHECK("k_synth_f32");
    return LLZ_OK;
}

extern "C" int llzs_synth_i16(short *dst, int channels, long n, long pitch, unsigned seed, int chan0, void *stream)
{
    if (!dst || channels <= 0 || n <= 0 || pitch < n || channels > 65535) {
        llzs_set_error("synth_i16: bad arguments");
        return LLZ_ERR_ARG;
    }
    dim3 grid((unsigned)((n + 1023) / 1024), (unsigned)channels);
    hipLaunchKernelGGL(k_synth_i16, grid, dim3(256), 0, as_stream(stream), dst, n, pitch, seed, chan0);
    LLZ_LAUNCH_CHECK("k_synth_i16");
    return LLZ_OK;
}

extern "C" int llz_hip_synth_f32(float *d, int c, long n, long p, unsigned s, int c0, void *st)
{
    return llzs_synth_f32(d, c, n, p, s, c0, st);
}
extern "C" int llz_hip_synth_i16(short *d, int c, long n, long p, unsigned s, int c0, void *st)
{
    return llzs_synth_i16(d, c, n, p, s, c0, st);
}
