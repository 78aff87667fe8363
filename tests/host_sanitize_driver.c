/* Drives the host C layer (csrc/host/\*.c) of libllzfilter_hip under AddressSanitizer + UBSan with the device shim stubbed
 * out (tests/test_host_sanitizers.py generates the stub from csrc/llz_shim.h: device memory is malloc, copies are memcpy,
 * kernels return LLZ_OK without computing).  What is checked is the handle layer's own memory handling -- table builders,
 * staging buffers, init / process / flush / uninit paths and their error exits -- not any arithmetic. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "llz_hip.h"
#include "llz_fir.h"
#include "llz_iir.h"
#include "llz_resample.h"
#include "llz_fft.h"
#include "llz_fft_fixed.h"
#include "llz_corr.h"
#include "llz_asmodel.h"
#include "llz_mdct.h"
#include "llz_mdct_fixed.h"
#include "llz_pcm.h"
#include "llz_shard.h"

#define BAD ((unsigned long)-1)
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "driver: %s failed at line %d (%s)\n", #c, __LINE__, llz_hip_last_error()); return 1; } } while (0)

int main(void)
{
    enum { N = 4096 };
    double *dx = calloc(N, sizeof(double)), *dy = calloc(N + 4096, sizeof(double));
    float *fx = calloc(16 * N, sizeof(float)), *fy = calloc(16 * N, sizeof(float));
    short *sx = calloc(16 * N, sizeof(short)), *sy = calloc(16 * N, sizeof(short));
    CHECK(dx && dy && fx && fy && sx && sy);

    /* tap design, windows (caller frees *h) */
    for (int win = 0; win < 3; win++) {
        double *h = NULL;
        CHECK(llz_fir_lpf_cof(&h, 63, 0.25, win) == 63); free(h);
        CHECK(llz_fir_hpf_cof(&h, 64, 0.25, win) == 65); free(h);
        CHECK(llz_fir_bandpass_cof(&h, 33, 0.1, 0.3, win) == 33); free(h);
        CHECK(llz_fir_bandstop_cof(&h, 16, 0.1, 0.3, win) == 17); free(h);
    }
    /* single-channel reference symbols */
    unsigned long h = llz_fir_filter_lpf_init(256, 63, 0.25, KAISER);
    CHECK(h != BAD);
    CHECK(llz_fir_filter(h, dx, dy, 256) == 256);
    CHECK(llz_fir_filter(h, dx, dy, 100) < 0);                 /* wrong frame length: refused */
    CHECK(llz_fir_filter_flush(h, dy) == 62);
    llz_fir_filter_uninit(h);
    h = llz_fir_filter_bandstop_init(128, 40, 0.1, 0.2, HAMMING); CHECK(h != BAD); llz_fir_filter_uninit(h);
    double a[3] = {1, -0.3, 0.2}, b[3] = {0.2, 0.4, 0.2};
    h = llz_iir_filter_init(2, a, 2, b); CHECK(h != BAD);
    CHECK(llz_iir_filter(h, dx, dy, 300) == 300);
    CHECK(llz_iir_filter_flush(h, dy) == 2);
    llz_iir_filter_uninit(h);
    /* batch FIR: every algorithm, host buffers (staging), flush, wrong sizes */
    const int algos[] = {LLZ_FIR_ALGO_AUTO, LLZ_FIR_ALGO_TIME, LLZ_FIR_ALGO_OVERLAP_SAVE, LLZ_FIR_ALGO_TIME_MFMA,
                         LLZ_FIR_ALGO_OVERLAP_SAVE_2048, LLZ_FIR_ALGO_OVERLAP_SAVE_4096};
    const int taps_n[] = {257, 63, 200, 300, 700, 2049};
    for (int i = 0; i < 6; i++) {
        float *t = calloc((size_t)taps_n[i], sizeof(float));
        h = llz_fir_filter_mc_init(4, 1000, t, taps_n[i], algos[i]);
        CHECK(h != BAD);
        CHECK(llz_fir_filter_mc(h, fx, fy, 1000) == 1000);
        CHECK(llz_fir_filter_mc(h, fx, fy, 999) < 0);
        CHECK(llz_fir_filter_mc_flush(h, fy) == taps_n[i] - 1);
        llz_fir_filter_mc_uninit(h);
        free(t);
    }
    h = llz_fir_filter_mc_lpf_init(2, 64, 33, 0.2, BLACKMAN); CHECK(h != BAD); llz_fir_filter_mc_uninit(h);
    CHECK(llz_fir_filter_mc_init(0, 64, fx, 3, 0) == BAD);
    CHECK(llz_fir_filter_mc_init(4, 64, fx, 300, LLZ_FIR_ALGO_OVERLAP_SAVE) == BAD);
    /* IIR cascade */
    double coef[3][6] = {{0.2, 0.4, 0.2, 1, -0.37, 0.2}, {0.01, 0, -0.01, 1, -1.9, 0.98}, {1, 0, 0, 1, 0, 0}};
    h = llz_iir_cascade_mc_init(5, 3, &coef[0][0]); CHECK(h != BAD);
    CHECK(llz_iir_cascade_mc(h, fx, fy, 3000) == 3000);
    CHECK(llz_iir_cascade_mc(h, fx, fy, 1) == 1);
    llz_iir_cascade_mc_uninit(h);
    CHECK(llz_iir_cascade_mc_init(5, 0, &coef[0][0]) == BAD);
    {   /* general direct form I for many channels: orders 3 / 3, streamed, flush; order 9 refused */
        double a3[4] = {1, -0.3695, 0.1958, 0}, b3[4] = {1.0, 0.2066, 0.4131, 0.2066};
        h = llz_iir_mc_init(6, 3, a3, 3, b3); CHECK(h != BAD);
        CHECK(llz_iir_mc(h, fx, fy, 2000) == 2000 && llz_iir_mc(h, fx, fy, 17) == 17);
        CHECK(llz_iir_mc_flush(h, fy) == 3);
        CHECK(llz_iir_mc(h, fx, fx, 10) < 0);
        llz_iir_mc_uninit(h);
        double a9[10] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0.1};
        CHECK(llz_iir_mc_init(2, 9, a9, 0, b3) == BAD);
        h = llz_iir_mc_init(2, 2, a3, 0, NULL); CHECK(h != BAD); CHECK(llz_iir_mc_flush(h, fy) == 0); llz_iir_mc_uninit(h);
    }
    {   /* eight high-Q sections: double arithmetic with the folded-gain tables of the 32-sample kernel; a zero b0: no folding */
        double hq[8][6];
        for (int s = 0; s < 8; s++) { hq[s][0] = 0.01; hq[s][1] = 0; hq[s][2] = -0.01; hq[s][3] = 1; hq[s][4] = -1.89 + 0.01 * s; hq[s][5] = 0.9801; }
        h = llz_iir_cascade_mc_init(4, 8, &hq[0][0]); CHECK(h != BAD);
        CHECK(llz_iir_cascade_mc_precision(h) == 64);
        CHECK(llz_iir_cascade_mc(h, fx, fy, 4096 + 1024 + 7) == 4096 + 1024 + 7);
        llz_iir_cascade_mc_uninit(h);
        hq[3][0] = 0.0;
        h = llz_iir_cascade_mc_init(4, 8, &hq[0][0]); CHECK(h != BAD);
        CHECK(llz_iir_cascade_mc(h, fx, fy, 2048) == 2048);
        llz_iir_cascade_mc_uninit(h);
    }
    /* resamplers: the reference's three and the batch forms (int16 screen tables, matrix-core band tables) */
    h = llz_resample_filter_init(147, 160, 1.0, BLACKMAN); CHECK(h != BAD);
    int nb = llz_get_resample_framelen_bytes(h), ob = 0;
    CHECK(nb > 0 && nb <= 2 * 16 * N);
    CHECK(llz_resample(h, (unsigned char *)sx, nb, (unsigned char *)sy, &ob) == 0 && ob > 0);
    CHECK(llz_resample(h, (unsigned char *)sx, nb - 2, (unsigned char *)sy, &ob) != 0);
    llz_resample_filter_uninit(h);
    h = llz_decimate_init(3, 1.0, HAMMING); CHECK(h != BAD);
    nb = llz_get_resample_framelen_bytes(h);
    CHECK(llz_decimate(h, (unsigned char *)sx, nb, (unsigned char *)sy, &ob) == 0);
    llz_decimate_uninit(h);
    h = llz_interp_init(2, 1.0, KAISER); CHECK(h != BAD);
    nb = llz_get_resample_framelen_bytes(h);
    CHECK(llz_interp(h, (unsigned char *)sx, nb, (unsigned char *)sy, &ob) == 0);
    llz_interp_uninit(h);
    CHECK(llz_resample_filter_init(1, 17, 1.0, HAMMING) == BAD);
    const int lm[][3] = {{1, 3, LLZ_PCM_F32}, {1, 3, LLZ_PCM_I16}, {1, 3, LLZ_PCM_I16_FAST}, {147, 160, LLZ_PCM_F32},
                         {160, 147, LLZ_PCM_I16}, {2, 3, LLZ_PCM_F32}, {441, 320, LLZ_PCM_F32}};
    for (int i = 0; i < 7; i++) {
        h = llz_resample_mc_init(3, lm[i][0], lm[i][1], 0.9, BLACKMAN, lm[i][2]);
        CHECK(h != BAD);
        const long n_in = (long)lm[i][1] * 8;
        CHECK(llz_resample_mc_out_len(h, n_in) == (long)lm[i][0] * 8);
        const void *in = lm[i][2] == LLZ_PCM_F32 ? (const void *)fx : (const void *)sx;
        void *out = lm[i][2] == LLZ_PCM_F32 ? (void *)fy : (void *)sy;
        CHECK(llz_resample_mc(h, in, n_in, out) == (long)lm[i][0] * 8);
        double *m = calloc((size_t)lm[i][0] * llz_resample_mc_sub_len(h), sizeof(double));
        CHECK(llz_resample_mc_get_matrix(h, m, lm[i][0] * llz_resample_mc_sub_len(h)) > 0);
        CHECK(llz_resample_mc_set_matrix(h, m, lm[i][0] * llz_resample_mc_sub_len(h)) == 0);
        free(m);
        llz_resample_mc_uninit(h);
    }
    /* transforms */
    h = llz_fft_init(1024); CHECK(h != BAD); llz_fft(h, dx); llz_ifft(h, dx); llz_fft_uninit(h);
    CHECK(llz_fft_init(48) == BAD);
    h = llz_fft_fixed_init(256); CHECK(h != BAD);
    llz_fft_fixed(h, (int *)fx); llz_ifft_fixed(h, (int *)fx);
    CHECK(llz_fft_fixed_batch(h, (int *)fx, 4) >= 0);
    llz_fft_fixed_uninit(h);
    h = llz_fft_batch_init(512); CHECK(h != BAD);
    CHECK(llz_fft_batch(h, fx, 8) >= 0 && llz_ifft_batch(h, fx, 8) >= 0);
    llz_fft_batch_uninit(h);
    /* correlation, frames, MDCT */
    llz_autocorr(dx, 512, 16, dy); llz_crosscorr(dx, dx, 512, 16, dy); (void)llz_corr_cof(dx, dx, 64);
    h = llz_autocorr_fast_init(300); CHECK(h != BAD); llz_autocorr_fast(h, dx, 300, 16, dy); llz_autocorr_fast_uninit(h);
    h = llz_autocorr_fast_mc_init(300, 16); CHECK(h != BAD); CHECK(llz_autocorr_fast_mc(h, fx, fy, 4) >= 0); llz_autocorr_fast_mc_uninit(h);
    h = llz_analysis_fft_init(0, 64, HAMMING); CHECK(h != BAD); llz_analysis_fft(h, dx, dy, dy + 1024); llz_analysis_fft_uninit(h);
    h = llz_synthesis_fft_init(1, 64, KAISER); CHECK(h != BAD); llz_synthesis_fft(h, dy, dy + 1024, dx); llz_synthesis_fft_uninit(h);
    h = llz_stft_mc_init(3, 0, 64, BLACKMAN); CHECK(h != BAD);
    CHECK(llz_stft_mc_analysis(h, fx, fy, fy + 8 * N, 5) >= 0);
    CHECK(llz_stft_mc_synthesis(h, fy, fy + 8 * N, fx, 5) >= 0);
    llz_stft_mc_uninit(h);
    for (int type = 0; type < 3; type++) {
        h = llz_mdct_init(type, 256); CHECK(h != BAD); llz_mdct(h, dx, dy); llz_imdct(h, dy, dx); llz_mdct_uninit(h);
        h = llz_mdct_fixed_init(type, 256); CHECK(h != BAD); llz_mdct_fixed(h, (int *)fx, (int *)fy); llz_imdct_fixed(h, (int *)fy, (int *)fx);
        CHECK(llz_mdct_fixed_len(h) == 256);
        CHECK(llz_mdct_fixed_batch(h, (int *)fx, (int *)fy, 5) == 5 && llz_imdct_fixed_batch(h, (int *)fy, (int *)fx, 5) == 5);
        CHECK(llz_mdct_fixed_batch(h, (int *)fx, (int *)fx, 1) < 0 && llz_mdct_fixed_batch(h, (int *)fx, (int *)fy, 0) < 0);
        llz_mdct_fixed_uninit(h);
    }
    h = llz_analysis_mdct_init(256, 0); CHECK(h != BAD); llz_analysis_mdct(h, dx, dy); llz_analysis_mdct_uninit(h);
    h = llz_synthesis_mdct_init(256, 1); CHECK(h != BAD); llz_synthesis_mdct(h, dy, dx); llz_synthesis_mdct_uninit(h);
    h = llz_mdct_batch_init(1024); CHECK(h != BAD); CHECK(llz_mdct_batch(h, fx, fy, 4) >= 0 && llz_imdct_batch(h, fy, fx, 4) >= 0);
    llz_mdct_batch_uninit(h);
    CHECK(llz_mdct_batch_init(100) == BAD);
    /* windowed MDCT frames in batch: both directions, two calls (state buffers swap), refused shapes */
    h = llz_mdct_frames_mc_init(3, 128, MDCT_KBD); CHECK(h != BAD);
    CHECK(llz_mdct_frames_mc_analysis(h, fx, fy, 5) == 5 && llz_mdct_frames_mc_synthesis(h, fy, fx, 5) == 5);
    CHECK(llz_mdct_frames_mc_analysis(h, fx, fy, 1) == 1 && llz_mdct_frames_mc_synthesis(h, fy, fx, 1) == 1);
    CHECK(llz_mdct_frames_mc_analysis(h, fx, fx, 1) < 0 && llz_mdct_frames_mc_analysis(h, fx, fy, 0) < 0);
    llz_mdct_frames_mc_uninit(h);
    CHECK(llz_mdct_frames_mc_init(3, 96, MDCT_SINE) == BAD && llz_mdct_frames_mc_init(0, 128, MDCT_SINE) == BAD);
    /* PCM, WAV */
    CHECK(llz_pcm_deinterleave_i16_f32(sx, fx, 6, 100, 1.0f / 32768, NULL) == 0);
    CHECK(llz_pcm_interleave_f32_i16(fx, sx, 6, 100, 32768.f, NULL) == 0);
    unsigned char wav[44 + 64];
    llz_wav_info wi = {1, 2, 48000, 2, 4, 16, 44};
    memset(wav, 0, sizeof wav);
    CHECK(llz_wav_write_header(wav, &wi) == 0);
    CHECK(llz_wav_parse(wav, sizeof wav, &wi) == 0 && wi.frames == 16 && wi.channels == 2);
    CHECK(llz_wav_ingest_f32(wav, sizeof wav, fy, 16, &wi, NULL) == 16);
    CHECK(llz_wav_ingest_f32(wav, 60, fy, 16, &wi, NULL) == 4);           /* truncated: 16 bytes of data left */
    CHECK(llz_wav_parse(wav, 20, &wi) < 0);
    /* sharded handles: three shards on "device" 0 */
    const int dev3[3] = {0, 0, 0};
    float taps33[33] = {0};
    h = llz_fir_filter_mc_sharded_init(7, 512, taps33, 33, 0, dev3, 3); CHECK(h != BAD);
    const float *ins[3]; float *outs[3];
    for (int s = 0; s < 3; s++) { int c0, cnt; CHECK(llz_sharded_shard(h, s, NULL, &c0, &cnt) == 0); ins[s] = fx + c0 * 512; outs[s] = fy + c0 * 512; }
    CHECK(llz_fir_filter_mc_sharded(h, ins, outs, 512) == 512);
    CHECK(llz_fir_filter_mc_sharded_flush(h, outs) == 32);
    CHECK(llz_sharded_synchronize(h) == 0);
    llz_sharded_uninit(h);
    h = llz_iir_cascade_mc_sharded_init(7, 3, &coef[0][0], dev3, 3); CHECK(h != BAD);
    CHECK(llz_iir_cascade_mc_sharded(h, ins, outs, 512) == 512);
    llz_sharded_uninit(h);
    h = llz_resample_mc_sharded_init(7, 1, 3, 1.0, BLACKMAN, LLZ_PCM_I16, dev3, 3); CHECK(h != BAD);
    llz_sharded_uninit(h);
    CHECK(llz_fir_filter_mc_sharded_init(2, 512, taps33, 33, 0, dev3, 3) == BAD);
    free(dx); free(dy); free(fx); free(fy); free(sx); free(sy);
    printf("HOST_SANITIZE_OK\n");
    return 0;
}
