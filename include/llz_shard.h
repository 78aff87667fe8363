/*
 * include/llz_shard.h -- one batch handle over several GPUs of a node, single process (SURVEY.md 8(b) "what the build
 * adds", 8(e)).
 *
 * In the reference one handle is one channel (libllzfilter/llz_fir.c:23-33, llz_iir.c:17-26, llz_resample.c:54-78), so
 * a batch splits into contiguous channel ranges with no data dependency between them.  A sharded handle owns one
 * multi-channel handle (include/llz_fir.h, llz_iir.h, llz_resample.h, Part 2) per entry of devices[], each with its own
 * stream and device-resident history / state:
 *   shard s of n owns channels [chan0_s, chan0_s + count_s): count = channels / n, the remainder spread over the first
 *   shards (llz_shard_range);
 *   the coefficient tables (taps, overlap-save spectrum and twiddles, biquad rows and transition powers, the L x Q
 *   resample matrix) are built ONCE, uploaded to devices[0] and broadcast to the other devices with one ncclBroadcast
 *   per table (RCCL over xGMI; librccl.so is loaded on first use, and not at all when every shard sits on one device);
 *   a call launches every shard's kernels back to back from the calling thread, each on its shard's stream: the GPUs run
 *   concurrently, there is no steady-state communication.
 * devices[] may name a device more than once (several shards on one GPU: that is how a one-GPU box runs the same code).
 *
 * Buffers are passed per shard: in[s] / out[s] is shard s's planar [count_s][frame_len] block, a device pointer on
 * devices[s] (used in place, asynchronously) or a host pointer (staged, synchronous).  A host caller holding one planar
 * [channels][frame_len] array passes in[s] = base + chan0_s * frame_len.
 * Errors: init returns (unsigned long)-1, calls a negative LLZ_ERR_* code; llz_hip_last_error() has the text.
 */
#ifndef LLZ_SHARD_H
#define LLZ_SHARD_H

#include "llz_hip.h"
#include "llz_fir.h"
#include "llz_iir.h"
#include "llz_resample.h"

#ifdef __cplusplus
extern "C" {
#endif

/* contiguous channel range of shard `shard` of `n_shards` */
int llz_shard_range(int channels, int n_shards, int shard, int *chan0, int *count);

/* ---- FIR: llz_fir_filter_mc_* (include/llz_fir.h) over devices[] ---- */
unsigned long llz_fir_filter_mc_sharded_init(int channels, int frame_len, const float *h, int flt_len, int algo,
                                             const int *devices, int n_devices);
int  llz_fir_filter_mc_sharded(unsigned long handle, const float *const *in, float *const *out, int frame_len);
int  llz_fir_filter_mc_sharded_flush(unsigned long handle, float *const *out);   /* out[s]: [count_s][flt_len-1] */

/* ---- IIR: llz_iir_cascade_mc_* (include/llz_iir.h) over devices[] ---- */
unsigned long llz_iir_cascade_mc_sharded_init(int channels, int stages, const double *coef, const int *devices,
                                              int n_devices);
int  llz_iir_cascade_mc_sharded(unsigned long handle, const float *const *x, float *const *y, int frame_len);

/* ---- resample: llz_resample_mc_* (include/llz_resample.h) over devices[] ---- */
unsigned long llz_resample_mc_sharded_init(int channels, int L, int M, double gain, win_t win_type, int pcm_format,
                                           const int *devices, int n_devices);
long llz_resample_mc_sharded(unsigned long handle, const void *const *in, long n_in, void *const *out);

/* ---- common to the three kinds ---- */
void llz_sharded_uninit(unsigned long handle);
int  llz_sharded_count(unsigned long handle);                                   /* number of shards */
int  llz_sharded_rccl_ranks(unsigned long handle);   /* ranks of the RCCL communicator that carried the coefficient tables at
                                                      * init: the number of distinct devices, 0 when all shards share one GPU */
int  llz_sharded_shard(unsigned long handle, int shard, int *device, int *chan0, int *count);
void *llz_sharded_stream(unsigned long handle, int shard);                      /* hipStream_t of a shard */
unsigned long llz_sharded_sub(unsigned long handle, int shard);                 /* the shard's own *_mc handle */
int  llz_sharded_synchronize(unsigned long handle);                             /* waits for every shard's stream */
/* per-GPU event timing: start / stop record one event on every shard's stream; the elapsed time of the
 * bracket is the MAX over shards, per_shard_ms (n_shards doubles or NULL) receives each shard's own */
int    llz_sharded_timer_start(unsigned long handle);
int    llz_sharded_timer_stop(unsigned long handle);
double llz_sharded_timer_ms(unsigned long handle, double *per_shard_ms);

#ifdef __cplusplus
}
#endif
#endif
