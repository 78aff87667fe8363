// common.hpp -- shared by the HIP translation units (gfx950 only)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../llz_shim.h"
#include "../../../include/llz_hip.h"

#define LLZ_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) {                                                         \
            llzs_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),       \
                           __FILE__, __LINE__);                                          \
            return LLZ_ERR_DEVICE;                                                       \
        }                                                                                \
    } while (0)

// kernel launches: catch configuration errors at the launch site
#define LLZ_LAUNCH_CHECK(name)                                                           \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            llzs_set_error("launch of %s failed: %s", name, hipGetErrorString(e__));     \
            return LLZ_ERR_DEVICE;                                                       \
        }                                                                                \
    } while (0)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// murmur3 finaliser; the counter hash of SURVEY.md section 8(d)
__host__ __device__ static inline uint32_t llz_fmix32(uint32_t u)
{
    u ^= u >> 16; u *= 0x85EBCA6Bu; u ^= u >> 13; u *= 0xC2B2AE35u; u ^= u >> 16;
    return u;
}
__host__ __device__ static inline uint32_t llz_synth_u32(uint32_t seed, uint32_t c, uint32_t n)
{
    return llz_fmix32(seed ^ (c * 0x9E3779B9u) ^ (n * 0x85EBCA6Bu));
}
