/* sharded_demo.c -- one batch handle over several GPUs from plain C (include/llz_shard.h).
 *
 *   ./sharded_demo [n_shards]          default: one shard per device of the node; on a one-GPU box pass e.g. 4 to run four
 *                                      shards on device 0 (the same code path)
 *
 * Filters 96 channels x 65536 samples of synthetic PCM (generated on each shard's device) with a 257-tap low-pass, once
 * through ONE unsharded handle on device 0 and once through a sharded handle, and requires identical output.  Then resamples
 * the same channels 3:1 to int16 through a sharded resampler (bit-exact path) and prints the per-GPU event times. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "llz_fir.h"
#include "llz_hip.h"
#include "llz_resample.h"
#include "llz_shard.h"

#define DIE(what) do { fprintf(stderr, "%s: %s\n", what, llz_hip_last_error()); return 1; } while (0)

int main(int argc, char **argv)
{
    enum { CH = 96, N = 65536, SEED = 7 };
    int ndev = llz_hip_device_count();
    if (ndev < 1) DIE("no device");
    int n_shards = argc > 1 ? atoi(argv[1]) : ndev;
    if (n_shards < 1 || n_shards > 64) n_shards = ndev;
    int devices[64];
    for (int s = 0; s < n_shards; s++) devices[s] = s % ndev;

    /* the unsharded result on device 0 */
    if (llz_hip_set_device(0) != LLZ_OK) DIE("set_device");
    float *x0 = llz_hip_malloc(sizeof(float) * CH * N), *y0 = llz_hip_malloc(sizeof(float) * CH * N);
    float *ref = malloc(sizeof(float) * CH * N), *got = malloc(sizeof(float) * CH * N);
    if (!x0 || !y0 || !ref || !got) DIE("alloc");
    if (llz_hip_synth_f32(x0, CH, N, N, SEED, 0, NULL) != LLZ_OK) DIE("synth");
    unsigned long one = llz_fir_filter_mc_lpf_init(CH, N, 257, 0.1, KAISER);
    if (one == LLZ_BAD_HANDLE) DIE("llz_fir_filter_mc_lpf_init");
    if (llz_fir_filter_mc(one, x0, y0, N) < 0) DIE("llz_fir_filter_mc");
    if (llz_hip_download(ref, y0, sizeof(float) * CH * N) != LLZ_OK) DIE("download");
    llz_fir_filter_mc_uninit(one);

    /* the same taps, designed by the same host code, for the sharded handle */
    double *h64 = NULL;
    const int T = llz_fir_lpf_cof(&h64, 257, 0.1, KAISER);
    float taps[257];
    for (int i = 0; i < T; i++) taps[i] = (float)h64[i];
    free(h64);
    unsigned long sh = llz_fir_filter_mc_sharded_init(CH, N, taps, T, LLZ_FIR_ALGO_AUTO, devices, n_shards);
    if (sh == LLZ_BAD_HANDLE) DIE("llz_fir_filter_mc_sharded_init");
    const float *in[64];
    float *out[64];
    for (int s = 0; s < n_shards; s++) {
        int dev, c0, cnt;
        llz_sharded_shard(sh, s, &dev, &c0, &cnt);
        if (llz_hip_set_device(dev) != LLZ_OK) DIE("set_device");
        float *xs = llz_hip_malloc(sizeof(float) * (size_t)cnt * N), *ys = llz_hip_malloc(sizeof(float) * (size_t)cnt * N);
        if (!xs || !ys) DIE("alloc shard");
        /* the shard's own channels: chan0 selects the rows of the synthetic stream */
        if (llz_hip_synth_f32(xs, cnt, N, N, SEED, c0, NULL) != LLZ_OK || llz_hip_synchronize(NULL) != LLZ_OK) DIE("synth shard");
        in[s] = xs; out[s] = ys;
    }
    llz_sharded_timer_start(sh);
    if (llz_fir_filter_mc_sharded(sh, in, out, N) < 0) DIE("llz_fir_filter_mc_sharded");
    llz_sharded_timer_stop(sh);
    double per[64];
    const double ms = llz_sharded_timer_ms(sh, per);
    if (llz_sharded_synchronize(sh) != LLZ_OK) DIE("sync");
    for (int s = 0; s < n_shards; s++) {
        int dev, c0, cnt;
        llz_sharded_shard(sh, s, &dev, &c0, &cnt);
        llz_hip_set_device(dev);
        if (llz_hip_download(got + (size_t)c0 * N, out[s], sizeof(float) * (size_t)cnt * N) != LLZ_OK) DIE("download shard");
    }
    if (memcmp(ref, got, sizeof(float) * CH * N) != 0) { fprintf(stderr, "sharded output differs from the unsharded one\n"); return 1; }
    printf("FIR 257 taps, %d ch x %d: %d shard(s) identical to one handle; %.3f ms (slowest shard)", CH, N, n_shards, ms);
    for (int s = 0; s < n_shards; s++) printf(" [%d: dev %d %.3f ms]", s, devices[s], per[s]);
    printf("\n");
    llz_sharded_uninit(sh);

    /* int16 3:1 decimation, sharded (tables: taps, fixed-point digit planes -- built once, broadcast) */
    unsigned long rs = llz_resample_mc_sharded_init(CH, 1, 3, 1.0, BLACKMAN, LLZ_PCM_I16, devices, n_shards);
    if (rs == LLZ_BAD_HANDLE) DIE("llz_resample_mc_sharded_init");
    const void *rin[64];
    void *rout[64];
    const long n_in = 3 * 8192;
    for (int s = 0; s < n_shards; s++) {
        int dev, c0, cnt;
        llz_sharded_shard(rs, s, &dev, &c0, &cnt);
        llz_hip_set_device(dev);
        short *xs = llz_hip_malloc(sizeof(short) * (size_t)cnt * n_in), *ys = llz_hip_malloc(sizeof(short) * (size_t)cnt * n_in / 3);
        if (!xs || !ys || llz_hip_synth_i16(xs, cnt, n_in, n_in, SEED, c0, NULL) != LLZ_OK || llz_hip_synchronize(NULL) != LLZ_OK)
            DIE("resample buffers");
        rin[s] = xs; rout[s] = ys;
    }
    if (llz_resample_mc_sharded(rs, rin, n_in, rout) != n_in / 3) DIE("llz_resample_mc_sharded");
    if (llz_sharded_synchronize(rs) != LLZ_OK) DIE("sync");
    printf("resample 1:3 int16, %d ch x %ld: %d shard(s)\n", CH, n_in, n_shards);
    llz_sharded_uninit(rs);
    for (int s = 0; s < n_shards; s++) {
        llz_hip_set_device(devices[s]);
        llz_hip_free((void *)in[s]); llz_hip_free(out[s]); llz_hip_free((void *)rin[s]); llz_hip_free(rout[s]);
    }
    llz_hip_set_device(0);
    llz_hip_free(x0); llz_hip_free(y0);
    free(ref); free(got);
    printf("OK\n");
    return 0;
}
