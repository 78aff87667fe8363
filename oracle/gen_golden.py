#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libllzref.so).

Run in the build container only (needs /root/reference to build _ref):  python oracle/gen_golden.py
The fixtures are data: seeded inputs and the outputs the reference's own C code produced for them
(gcc 11.4 -O2, glibc 2.35).  The reference has no golden vectors of its own, so these
are the pins for the oracle on the GPU box where the reference does not exist.

TEST INFRASTRUCTURE ONLY.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    po.build()
    r = po.Ref()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261004)

    # ---- design: 4 filter types x 3 windows x odd/even N, windows, estimators ----------------
    d = {}
    for kind in range(4):
        for win in range(3):
            for n in (15, 16, 63, 64, 257):
                d[f"taps_k{kind}_w{win}_n{n}"] = r.fir_design(kind, n, 0.2 if kind >= 2 else 0.25, 0.4, win)
    d["taps_lpf_kaiser_257_fc0p1"] = r.fir_design(po.LPF, 257, 0.1, 0.0, po.KAISER)      # bench config 3 taps
    d["taps_lpf_hamming_63_fc0p25"] = r.fir_design(po.LPF, 63, 0.25, 0.0, po.HAMMING)    # bench config 2 taps
    for win in range(3):
        for n in (8, 33):
            d[f"win_w{win}_n{n}"] = r.window(win, n)
    d["win_kaiser_beta5_n21"] = r.window(po.KAISER, 21, beta=5.0)
    ft = np.array([0.05, 0.15, 0.15 / 3, 0.15 / 160, 0.3])
    d["cofnum_ft"] = ft
    d["cofnum_hamming"] = np.array([r.cof_num(0, f) for f in ft])
    d["cofnum_blackman"] = np.array([r.cof_num(1, f) for f in ft])
    d["cofnum_kaiser90"] = np.array([r.cof_num(2, f, 90.0) for f in ft])
    d["cofnum_kaiser20"] = np.array([r.cof_num(2, f, 20.0) for f in ft])
    at = np.array([10.0, 21.0, 30.0, 40.0, 49.9, 50.0, 60.0, 90.0])
    d["atten"] = at
    d["atten2beta"] = np.array([r.atten2beta(a) for a in at])
    np.savez_compressed(os.path.join(OUT, "design.npz"), **d)

    # ---- streaming FIR (equal frame lengths only: SURVEY.md M8) + flush ----------------------
    d = {}
    x = rng.standard_normal(64 * 6)
    d["x"] = x
    for kind in range(4):
        for win in range(3):
            y, tail, taps = r.fir_stream(kind, 64, 31, 0.2 if kind >= 2 else 0.3, 0.45, win, x)
            d[f"y_k{kind}_w{win}"] = y
            d[f"tail_k{kind}_w{win}"] = tail
    # float32-representable input through a 257-tap Kaiser LPF, frame 512 (what the GPU batch path sees)
    x32 = rng.uniform(-1, 1, 512 * 3).astype(np.float32)
    y, tail, _ = r.fir_stream(po.LPF, 512, 257, 0.1, 0.0, po.KAISER, x32.astype(np.float64))
    d["x32_257"] = x32
    d["y32_257"] = y
    d["tail32_257"] = tail
    x32 = rng.uniform(-1, 1, 256 * 4).astype(np.float32)
    y, tail, _ = r.fir_stream(po.LPF, 256, 63, 0.25, 0.0, po.HAMMING, x32.astype(np.float64))
    d["x32_63"] = x32
    d["y32_63"] = y
    d["tail32_63"] = tail
    # impulse -> taps
    imp = np.zeros(16)
    imp[0] = 1.0
    y, _, taps = r.fir_stream(po.LPF, 16, 7, 0.3, 0.0, po.BLACKMAN, imp, flush=False)
    d["impulse_y"] = y
    d["impulse_taps"] = taps
    np.savez_compressed(os.path.join(OUT, "fir_stream.npz"), **d)

    # ---- IIR ----------------------------------------------------------------------------------
    d = {}
    a2 = np.array([1.0, -0.3695, 0.1958])
    b2 = np.array([0.2066, 0.4131, 0.2066])
    x = rng.standard_normal(300)
    d["x"] = x
    d["a2"], d["b2"] = a2, b2
    d["y2"], d["tail2"] = r.iir_stream(a2, b2, x, frame_len=100)
    a3 = np.array([1.0, -0.3695, 0.1958, 0.0])             # the order-3 call of llz_musicpitch.c:1277-1285
    b3 = np.array([1.0, 0.2066, 0.4131, 0.2066])
    d["a3"], d["b3"] = a3, b3
    d["y3"], d["tail3"] = r.iir_stream(a3, b3, x, frame_len=75)
    a5 = np.array([1.0, -1.2, 0.9, -0.3, 0.05, -0.01])     # M != N
    b5 = np.array([0.5, -0.25, 0.125])
    d["a5"], d["b5"] = a5, b5
    d["y5"], d["tail5"] = r.iir_stream(a5, b5, x, frame_len=300)
    # high-Q biquad (pole radius 0.99) on float32-representable input
    th = 0.3
    aq = np.array([1.0, -2 * 0.99 * np.cos(th), 0.99 ** 2])
    bq = np.array([0.01, 0.0, -0.01])
    x32 = rng.uniform(-1, 1, 2048).astype(np.float32)
    d["aq"], d["bq"], d["x32"] = aq, bq, x32
    d["yq"], _ = r.iir_stream(aq, bq, x32.astype(np.float64), flush=False)
    # 8-stage cascade = 8 chained handles, bench config 4 coefficients
    t = x32.astype(np.float64)
    for _ in range(8):
        t, _ = r.iir_stream(a2, b2, t, flush=False)
    d["y_cascade8"] = t
    imp = np.zeros(4)
    imp[0] = 1
    d["impulse2"], _ = r.iir_stream(a2, b2, imp, flush=False)
    np.savez_compressed(os.path.join(OUT, "iir.npz"), **d)

    # ---- resample family ------------------------------------------------------------------------
    d = {}
    for (L, M) in [(1, 3), (2, 3), (3, 2), (147, 160), (160, 147)]:
        for win in range(3):
            nin = r.rs_bytes_in(2, L, M, 1.0, win) // 2
            frames = 3 if nin < 4000 else 2
            seed = 1000 + 17 * L + M + win
            pcm = np.random.default_rng(seed).integers(-16384, 16384, nin * frames).astype(np.int16)
            y = r.rs_stream(2, L, M, 1.0, win, pcm)
            d[f"rs_{L}_{M}_w{win}_seed"] = np.array([seed, nin, frames])
            d[f"rs_{L}_{M}_w{win}_out"] = y
    # clipping case: full-scale square wave, gain 2 -> clamps both rails
    nin = r.rs_bytes_in(2, 2, 3, 2.0, po.BLACKMAN) // 2
    sq = np.where((np.arange(nin * 2) // 37) % 2 == 0, 32767, -32768).astype(np.int16)
    d["rs_clip_in"] = sq
    d["rs_clip_out"] = r.rs_stream(2, 2, 3, 2.0, po.BLACKMAN, sq)
    # impulse 1:3
    nin = r.rs_bytes_in(2, 1, 3, 1.0, po.BLACKMAN) // 2
    imp = np.zeros(nin, dtype=np.int16)
    imp[0] = 32767
    d["rs_impulse_out"] = r.rs_stream(2, 1, 3, 1.0, po.BLACKMAN, imp)
    d["rs_refused"] = np.array([r.rs_bytes_in(2, 17, 1, 1.0, 1) is None, r.rs_bytes_in(2, 1, 17, 1.0, 1) is None,
                                r.rs_bytes_in(2, 16, 1, 1.0, 1) is None])
    for M in (2, 3, 5):
        nin = r.rs_bytes_in(0, 1, M, 1.0, po.BLACKMAN) // 2
        pcm = np.random.default_rng(2000 + M).integers(-16384, 16384, nin * 3).astype(np.int16)
        d[f"dec_{M}_seed"] = np.array([2000 + M, nin, 3])
        d[f"dec_{M}_out"] = r.rs_stream(0, 1, M, 1.0, po.BLACKMAN, pcm)
    for L in (2, 3):
        nin = r.rs_bytes_in(1, L, 1, 1.0, po.BLACKMAN) // 2
        pcm = np.random.default_rng(3000 + L).integers(-16384, 16384, nin * 2).astype(np.int16)
        d[f"int_{L}_seed"] = np.array([3000 + L, nin, 2])
        # zeros behind each frame make the reference's read-past-the-end defined
        d[f"int_{L}_out"] = r.rs_stream(1, L, 1, 1.0, po.BLACKMAN, pcm, pad_tail=1024)
    d["bytes_in"] = np.array([r.rs_bytes_in(2, 1, 3, 1.0, 1), r.rs_bytes_in(2, 2, 3, 1.0, 1),
                              r.rs_bytes_in(2, 3, 2, 1.0, 1), r.rs_bytes_in(2, 147, 160, 1.0, 1),
                              r.rs_bytes_in(2, 160, 147, 1.0, 1), r.rs_bytes_in(0, 1, 3, 1.0, 1),
                              r.rs_bytes_in(1, 3, 1, 1.0, 1)])
    np.savez_compressed(os.path.join(OUT, "resample.npz"), **d)

    # ---- FFTs -------------------------------------------------------------------------------------
    d = {}
    for n in (8, 64, 1024, 4096):
        g = np.random.default_rng(4000 + n)
        z = (g.uniform(-1, 1, n) + 1j * g.uniform(-1, 1, n)).astype(np.complex64).astype(np.complex128)
        d[f"fft_in_{n}"] = z.astype(np.complex64)
        d[f"fft_fwd_{n}"] = r.fft(z)
        d[f"fft_inv_{n}"] = r.fft(z, inverse=True)
        q = g.integers(-10000, 10001, 2 * n).astype(np.int32)
        d[f"fftx_in_{n}"] = q
        d[f"fftx_fwd_{n}"] = r.fft_fixed(q)
        d[f"fftx_inv_{n}"] = r.fft_fixed(q, inverse=True)
    ramp = np.zeros(16, dtype=np.int32)
    ramp[0::2] = 1000 * np.arange(8)
    d["fftx_ramp_fwd"] = r.fft_fixed(ramp)
    d["fftx_ramp_roundtrip"] = r.fft_fixed(r.fft_fixed(ramp), inverse=True)
    s = np.zeros(2048, dtype=np.int32)
    s[0::2] = (10000 * np.sin(0.3 * np.arange(1024))).astype(np.int32)
    d["fftx_sin_in"] = s
    d["fftx_sin_fwd"] = r.fft_fixed(s)
    d["fftx_sin_roundtrip"] = r.fft_fixed(r.fft_fixed(s), inverse=True)
    np.savez_compressed(os.path.join(OUT, "fft.npz"), **d)

    # ---- correlation ----------------------------------------------------------
    d = {}
    g = np.random.default_rng(5000)
    for n, p in ((64, 10), (300, 16), (1024, 32), (2048, 2047)):
        x = g.uniform(-1, 1, n).astype(np.float32).astype(np.float64)
        y = g.uniform(-1, 1, n).astype(np.float32).astype(np.float64)
        d[f"x_{n}"], d[f"y_{n}"] = x, y
        d[f"auto_{n}_{p}"] = r.autocorr(x, p)
        d[f"cross_{n}_{p}"] = r.crosscorr(x, y, p)
        d[f"fast_{n}_{p}"] = r.autocorr_fast(x, p)
        d[f"cof_{n}"] = np.array([r.corr_cof(x, y)])
    np.savez_compressed(os.path.join(OUT, "corr.npz"), **d)

    # ---- windowed-FFT analysis / synthesis frames -----------------------------
    d = {}
    g = np.random.default_rng(6000)
    for hint, frame_len, win, frames in ((0, 8, po.HAMMING, 9), (0, 64, po.BLACKMAN, 7), (0, 256, po.KAISER, 6),
                                         (1, 8, po.KAISER, 9), (1, 128, po.HAMMING, 5), (1, 512, po.BLACKMAN, 4)):
        x = g.uniform(-1, 1, frame_len * frames).astype(np.float32).astype(np.float64)
        re, im = r.stft_analysis(hint, frame_len, win, x)
        key = f"{hint}_{frame_len}_{win}"
        d["x_" + key], d["re_" + key], d["im_" + key] = x, re, im
        d["syn_" + key] = r.stft_synthesis(hint, frame_len, win, re, im)
    np.savez_compressed(os.path.join(OUT, "stft.npz"), **d)

    # ---- MDCT -----------------------------------------------------------------
    d = {}
    g = np.random.default_rng(7000)
    for n in (16, 64, 256, 2048):
        d[f"sine_{n}"] = r.mdct_window(0, n)
        d[f"kbd_{n}"] = r.mdct_window(1, n)
        x = g.uniform(-1, 1, n).astype(np.float32).astype(np.float64)
        d[f"x_{n}"] = x
        for t in (0, 1, 2):
            if t == 0 and n > 256:
                continue
            X = r.mdct(t, x)
            d[f"mdct{t}_{n}"] = X
            d[f"imdct{t}_{n}"] = r.imdct(t, X)
    for n in (16, 64, 256, 2048):
        xi = g.integers(-20000, 20001, n).astype(np.int32)
        d[f"xi_{n}"] = xi
        for t in (0, 1, 2):
            if t == 0 and n > 256:
                continue
            Xi = r.mdct_fixed(t, xi)
            d[f"mdctx{t}_{n}"] = Xi
            d[f"imdctx{t}_{n}"] = r.mdct_fixed(t, Xi, inverse=True)
    for frame_len, win in ((8, 0), (64, 1), (512, 0)):
        x = g.uniform(-1, 1, frame_len * 6).astype(np.float32).astype(np.float64)
        X, y = r.mdct_frames(frame_len, win, x)
        d[f"fx_{frame_len}_{win}"], d[f"fX_{frame_len}_{win}"], d[f"fy_{frame_len}_{win}"] = x, X, y
    np.savez_compressed(os.path.join(OUT, "mdct.npz"), **d)

    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden fixtures written to", OUT, "total bytes", tot)


if __name__ == "__main__":
    main()
