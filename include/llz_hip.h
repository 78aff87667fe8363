/*
 * include/llz_hip.h -- common part of the libllzfilter_hip.so C ABI: error codes, device selection,
 * device buffers and the synthetic-PCM generator used by the benchmark.
 * Plain C: pointers and sizes only. A hipStream_t crosses the boundary as void*.
 */
#ifndef LLZ_HIP_H
#define LLZ_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLZ_OK            0
#define LLZ_ERR_ARG     (-1)   /* bad argument (size mismatch, NULL, unsupported length) */
#define LLZ_ERR_DEVICE  (-2)   /* HIP runtime error; llz_hip_last_error() has the text */
#define LLZ_ERR_NOMEM   (-3)
#define LLZ_ERR_RANGE   (-4)   /* configuration outside what the kernels support */
#define LLZ_BAD_HANDLE  ((unsigned long)-1)   /* same failure value as llz_resample.c:278-279 */

const char *llz_hip_last_error(void);          /* thread-local text of the last failure */
int   llz_hip_device_count(void);              /* < 0 when no HIP device / runtime */
int   llz_hip_set_device(int device);
int   llz_hip_get_device(void);
int   llz_hip_synchronize(void *stream);       /* hipStreamSynchronize (NULL = default stream) */

/* Measurement / test override of one of the library's own choices (kernel form, grid shape): name as listed by
 * llz_hip_tune_name(i), value < 0 clears it.  Returns LLZ_OK, or LLZ_ERR_ARG for an unknown name.  The
 * library never reads environment variables; without calls to this function every choice follows from the problem. */
int         llz_hip_tune(const char *name, int value);
const char *llz_hip_tune_name(int index);      /* NULL past the last name */

void *llz_hip_malloc(size_t bytes);            /* device memory, NULL on failure */
void  llz_hip_free(void *dev_ptr);
int   llz_hip_upload(void *dev_dst, const void *host_src, size_t bytes);
int   llz_hip_download(void *host_dst, const void *dev_src, size_t bytes);
int   llz_hip_is_device_ptr(const void *p);    /* 1: memory of the CURRENT device; 0: host memory; LLZ_ERR_ARG: memory of
                                                * another GPU (every batch call refuses such a buffer the same way) */

/* counter-hash PCM written straight into device memory, planar [channels][stride]:
 *   u = fmix32(seed ^ (chan0+c)*0x9E3779B9 ^ n*0x85EBCA6B)
 *   f32: (float)(u>>8) * 2^-23 - 1  in [-1,1);   i16: (u>>17) - 16384 in [-16384,16383] */
int llz_hip_synth_f32(float *dev_dst, int channels, long n, long stride, unsigned seed, int chan0, void *stream);
int llz_hip_synth_i16(short *dev_dst, int channels, long n, long stride, unsigned seed, int chan0, void *stream);

/* event timing of work on a stream, measured on THAT stream: t = llz_hip_timer_new(); start; ...; stop -> ms */
void  *llz_hip_timer_new(void);
int    llz_hip_timer_start(void *timer, void *stream);
int    llz_hip_timer_stop(void *timer, void *stream);
double llz_hip_timer_ms(void *timer);          /* synchronises on the stop event */
void   llz_hip_timer_free(void *timer);

#ifdef __cplusplus
}
#endif
#endif
