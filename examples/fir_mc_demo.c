/*
 * examples/fir_mc_demo.c -- a plain C caller of libllzfilter_hip.so through the public headers only.
 *
 *   gcc -O2 -Iinclude examples/fir_mc_demo.c -Lllzlab_amd -lllzfilter_hip -Wl,-rpath,$PWD/llzlab_amd -lm -o fir_mc_demo
 *
 * Part 1 uses the reference's own single-channel API exactly as a libllzfilter caller would
 * (reference libllzfilter/llz_fir.h:38-58): nothing in the calling code knows a GPU is involved.
 * Part 2 uses the multi-channel batch extension on host buffers and checks it against part 1 channel by channel.
 * Prints "OK" and exits 0 when the two agree to 1e-5 RMS.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "llz_fir.h"
#include "llz_hip.h"

#define CHANNELS 4
#define FRAME    2048
#define TAPS     63

int main(void)
{
    if (llz_hip_device_count() < 1) {
        fprintf(stderr, "no GPU: %s\n", llz_hip_last_error());
        return 2;
    }
    static float x[CHANNELS][FRAME], y[CHANNELS][FRAME];
    static double xd[FRAME], yd[CHANNELS][FRAME];
    unsigned seed = 12345u;
    for (int c = 0; c < CHANNELS; c++)
        for (int i = 0; i < FRAME; i++) {
            seed = seed * 1664525u + 1013904223u;
            x[c][i] = (float)(seed >> 8) / 8388608.0f - 1.0f;
        }

    /* part 1: the reference API, one handle per channel */
    for (int c = 0; c < CHANNELS; c++) {
        unsigned long h = llz_fir_filter_lpf_init(FRAME, TAPS, 0.25, HAMMING);
        if (h == (unsigned long)-1) { fprintf(stderr, "init: %s\n", llz_hip_last_error()); return 1; }
        for (int i = 0; i < FRAME; i++) xd[i] = x[c][i];
        if (llz_fir_filter(h, xd, yd[c], FRAME) != FRAME) { fprintf(stderr, "%s\n", llz_hip_last_error()); return 1; }
        llz_fir_filter_uninit(h);
    }

    /* part 2: all channels in one call */
    unsigned long hm = llz_fir_filter_mc_lpf_init(CHANNELS, FRAME, TAPS, 0.25, HAMMING);
    if (hm == LLZ_BAD_HANDLE) { fprintf(stderr, "mc init: %s\n", llz_hip_last_error()); return 1; }
    if (llz_fir_filter_mc(hm, &x[0][0], &y[0][0], FRAME) != FRAME) { fprintf(stderr, "%s\n", llz_hip_last_error()); return 1; }
    llz_fir_filter_mc_uninit(hm);

    double err = 0.0, ref = 0.0;
    for (int c = 0; c < CHANNELS; c++)
        for (int i = 0; i < FRAME; i++) {
            const double d = (double)y[c][i] - yd[c][i];
            err += d * d;
            ref += yd[c][i] * yd[c][i];
        }
    err = sqrt(err / (CHANNELS * FRAME));
    ref = sqrt(ref / (CHANNELS * FRAME));
    printf("rms error %.3g (signal rms %.3g)\n", err, ref);
    if (!(err <= 1e-5) || !(err / ref <= 1e-5)) { printf("MISMATCH\n"); return 1; }
    printf("OK\n");
    return 0;
}
