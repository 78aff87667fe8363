"""Pin the oracle (oracle/llz_oracle.c) against the committed golden fixtures.

The fixtures in tests/golden/ were produced by the reference's own C files compiled in the build container
(oracle/gen_golden.py, oracle/_ref).  Everything here is bit-exact: double, int16 and int32 alike.
Runs without a GPU and without /root/reference.
"""
import os

import numpy as np
import pytest

from oracle import pyoracle as po

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def same(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)


def test_design_taps_all_types_windows(oracle):
    d = load("design.npz")
    for kind in range(4):
        for win in range(3):
            for n in (15, 16, 63, 64, 257):
                got = oracle.fir_design(kind, n, 0.2 if kind >= 2 else 0.25, 0.4, win)
                assert same(got, d[f"taps_k{kind}_w{win}_n{n}"]), (kind, win, n)
    assert same(oracle.fir_design(po.LPF, 257, 0.1, 0.0, po.KAISER), d["taps_lpf_kaiser_257_fc0p1"])
    assert same(oracle.fir_design(po.LPF, 63, 0.25, 0.0, po.HAMMING), d["taps_lpf_hamming_63_fc0p25"])


def test_design_known_answers(oracle):
    # SURVEY.md section 4 table, captured from the compiled reference
    h = oracle.fir_design(po.LPF, 63, 0.25, 0.0, po.HAMMING)
    assert h[0] == -0.0005808492362303913 and h[31] == 0.25 and h[30] == 0.22454786737500823
    h = oracle.fir_design(po.LPF, 257, 0.1, 0.0, po.KAISER)
    assert h[0] == 1.3879661349850661e-06 and h[128] == 0.1
    h = oracle.fir_design(po.HPF, 64, 0.25, 0.0, po.HAMMING)
    assert len(h) == 65 and h[32] == 0.75                       # even N forced odd
    h = oracle.fir_design(po.LPF, 64, 0.25, 0.0, po.BLACKMAN)
    assert len(h) == 64 and h[31] == h[32] == 0.24337556806149604
    assert oracle.fir_design(po.BPF, 63, 0.2, 0.4, po.HAMMING)[31] == 0.2
    assert oracle.fir_design(po.BSF, 63, 0.2, 0.4, po.HAMMING)[31] == 0.8


def test_windows_and_estimators(oracle):
    d = load("design.npz")
    for win in range(3):
        for n in (8, 33):
            assert same(oracle.window(win, n), d[f"win_w{win}_n{n}"])
    assert same(oracle.window(po.KAISER, 21, beta=5.0), d["win_kaiser_beta5_n21"])
    ft = d["cofnum_ft"]
    assert same(np.array([oracle.cof_num(0, f) for f in ft]), d["cofnum_hamming"])
    assert same(np.array([oracle.cof_num(1, f) for f in ft]), d["cofnum_blackman"])
    assert same(np.array([oracle.cof_num(2, f, 90.0) for f in ft]), d["cofnum_kaiser90"])
    assert same(np.array([oracle.cof_num(2, f, 20.0) for f in ft]), d["cofnum_kaiser20"])
    assert same(np.array([oracle.atten2beta(a) for a in d["atten"]]), d["atten2beta"])
    assert (oracle.cof_num(0, 0.05), oracle.cof_num(1, 0.05), oracle.cof_num(2, 0.05)) == (124, 131, 228)   # 6.6/0.05 = 131.99999999999997 in double


def test_fir_stream_and_flush(oracle):
    d = load("fir_stream.npz")
    for kind in range(4):
        for win in range(3):
            y, tail, _ = oracle.fir_stream(kind, 64, 31, 0.2 if kind >= 2 else 0.3, 0.45, win, d["x"])
            assert same(y, d[f"y_k{kind}_w{win}"]) and same(tail, d[f"tail_k{kind}_w{win}"]), (kind, win)
    y, tail, _ = oracle.fir_stream(po.LPF, 512, 257, 0.1, 0.0, po.KAISER, d["x32_257"].astype(np.float64))
    assert same(y, d["y32_257"]) and same(tail, d["tail32_257"])
    y, tail, _ = oracle.fir_stream(po.LPF, 256, 63, 0.25, 0.0, po.HAMMING, d["x32_63"].astype(np.float64))
    assert same(y, d["y32_63"]) and same(tail, d["tail32_63"])


def test_fir_impulse_gives_taps(oracle):
    d = load("fir_stream.npz")
    imp = np.zeros(16)
    imp[0] = 1.0
    y, _, taps = oracle.fir_stream(po.LPF, 16, 7, 0.3, 0.0, po.BLACKMAN, imp, flush=False)
    assert same(y, d["impulse_y"]) and same(taps, d["impulse_taps"])
    assert np.array_equal(y[:7], taps)


def test_fir_batch_driver_equals_stream(oracle):
    d = load("fir_stream.npz")
    h = oracle.fir_design(po.LPF, 257, 0.1, 0.0, po.KAISER)
    x = d["x32_257"].reshape(1, -1)
    # one frame of the whole length == three frames of 512 (equal-frame streaming is frame-split invariant)
    assert same(oracle.fir_batch_f32(x, h)[0], d["y32_257"])


def test_fir_oversize_frame_refused(oracle):
    f = oracle.lib.orc_fir_new(0, 8, 5, 0.3, 0.0, 0)
    x = np.zeros(16)
    assert oracle.lib.orc_fir_run(f, po._ptr(x), po._ptr(x.copy()), 16) == -1
    oracle.lib.orc_fir_free(f)


def test_iir(oracle):
    d = load("iir.npz")
    y, t = oracle.iir_stream(d["a2"], d["b2"], d["x"], frame_len=100)
    assert same(y, d["y2"]) and same(t, d["tail2"])
    y, t = oracle.iir_stream(d["a3"], d["b3"], d["x"], frame_len=75)
    assert same(y, d["y3"]) and same(t, d["tail3"])
    y, t = oracle.iir_stream(d["a5"], d["b5"], d["x"], frame_len=300)
    assert same(y, d["y5"]) and same(t, d["tail5"])
    y, _ = oracle.iir_stream(d["aq"], d["bq"], d["x32"].astype(np.float64), flush=False)
    assert same(y, d["yq"])
    imp = np.zeros(4)
    imp[0] = 1
    y, _ = oracle.iir_stream(d["a2"], d["b2"], imp, flush=False)
    assert same(y, d["impulse2"])
    assert list(y) == [0.2066, 0.4894387, 0.34699531964999997, 0.032382673150674987]


def test_iir_cascade_batch_driver(oracle):
    d = load("iir.npz")
    coef = np.tile(np.concatenate([d["b2"], d["a2"]]), (8, 1))
    y = oracle.iir_cascade_batch_f32(d["x32"].reshape(1, -1), coef)
    assert same(y[0], d["y_cascade8"])


def test_resample_rational(oracle):
    d = load("resample.npz")
    for (L, M) in [(1, 3), (2, 3), (3, 2), (147, 160), (160, 147)]:
        for win in range(3):
            seed, nin, frames = (int(v) for v in d[f"rs_{L}_{M}_w{win}_seed"])
            assert oracle.rs_info(2, L, M, 1.0, win)["bytes_in"] == 2 * nin
            pcm = np.random.default_rng(seed).integers(-16384, 16384, nin * frames).astype(np.int16)
            assert same(oracle.rs_stream(2, L, M, 1.0, win, pcm), d[f"rs_{L}_{M}_w{win}_out"]), (L, M, win)


def test_resample_clip_impulse_refusal(oracle):
    d = load("resample.npz")
    out = oracle.rs_stream(2, 2, 3, 2.0, po.BLACKMAN, d["rs_clip_in"])
    assert same(out, d["rs_clip_out"])
    assert out.max() == 32767 and out.min() == -32768           # both rails reached
    nin = oracle.rs_info(2, 1, 3, 1.0, po.BLACKMAN)["bytes_in"] // 2
    imp = np.zeros(nin, dtype=np.int16)
    imp[0] = 32767
    out = oracle.rs_stream(2, 1, 3, 1.0, po.BLACKMAN, imp)
    assert same(out, d["rs_impulse_out"])
    assert out[22] == 10922 and not out[:22].any() and not out[23:25].any()
    refused = [oracle.rs_info(2, 17, 1, 1.0, 1) is None, oracle.rs_info(2, 1, 17, 1.0, 1) is None,
               oracle.rs_info(2, 16, 1, 1.0, 1) is None]
    assert refused == [bool(v) for v in d["rs_refused"]] == [True, True, False]


def test_decimate_interp_and_frame_sizes(oracle):
    d = load("resample.npz")
    for M in (2, 3, 5):
        seed, nin, frames = (int(v) for v in d[f"dec_{M}_seed"])
        pcm = np.random.default_rng(seed).integers(-16384, 16384, nin * frames).astype(np.int16)
        assert same(oracle.rs_stream(0, 1, M, 1.0, po.BLACKMAN, pcm), d[f"dec_{M}_out"])
    for L in (2, 3):
        seed, nin, frames = (int(v) for v in d[f"int_{L}_seed"])
        pcm = np.random.default_rng(seed).integers(-16384, 16384, nin * frames).astype(np.int16)
        assert same(oracle.rs_stream(1, L, 1, 1.0, po.BLACKMAN, pcm), d[f"int_{L}_out"])
    got = [oracle.rs_info(2, 1, 3, 1.0, 1)["bytes_in"], oracle.rs_info(2, 2, 3, 1.0, 1)["bytes_in"],
           oracle.rs_info(2, 3, 2, 1.0, 1)["bytes_in"], oracle.rs_info(2, 147, 160, 1.0, 1)["bytes_in"],
           oracle.rs_info(2, 160, 147, 1.0, 1)["bytes_in"], oracle.rs_info(0, 1, 3, 1.0, 1)["bytes_in"],
           oracle.rs_info(1, 3, 1, 1.0, 1)["bytes_in"]]
    assert got == list(d["bytes_in"]) == [3072, 3072, 3072, 47040, 47040, 2046, 2048]
    info = oracle.rs_info(2, 1, 3, 1.0, po.BLACKMAN)
    assert (info["n"], info["cols"]) == (133, 134)
    info = oracle.rs_info(2, 147, 160, 1.0, po.BLACKMAN)
    assert (info["n"], info["cols"]) == (6763, 47)


def test_resample_batch_drivers(oracle):
    # int16 batch == per-channel streaming; float batch == int16 path before clamp/trunc on integer input
    x = oracle.synth_i16(3, 1536 * 4, seed=5)
    yb = oracle.rs_batch_i16(x, 1, 3, 1.0, po.BLACKMAN)
    for c in range(3):
        assert same(yb[c], oracle.rs_stream(2, 1, 3, 1.0, po.BLACKMAN, x[c]))
    yf = oracle.rs_batch_f32(x.astype(np.float32), 1, 3, 1.0, po.BLACKMAN)
    assert np.array_equal(np.trunc(np.clip(yf, -32768, 32767)).astype(np.int16), yb)
    x = oracle.synth_i16(2, 3072 * 2, seed=6)
    yb = oracle.rs_batch_i16(x, 2, 3, 1.0, po.HAMMING)
    yf = oracle.rs_batch_f32(x.astype(np.float32), 2, 3, 1.0, po.HAMMING)
    assert np.array_equal(np.trunc(np.clip(yf, -32768, 32767)).astype(np.int16), yb)


@pytest.mark.parametrize("n", [8, 64, 1024, 4096])
def test_fft_float(oracle, n):
    d = load("fft.npz")
    z = d[f"fft_in_{n}"].astype(np.complex128)
    assert same(oracle.fft(z), d[f"fft_fwd_{n}"])
    assert same(oracle.fft(z, inverse=True), d[f"fft_inv_{n}"])
    # forward is unscaled, inverse divides by N (the reference's header comments claim the opposite)
    assert np.allclose(oracle.fft(z), np.fft.fft(z), rtol=0, atol=1e-9 * n)
    assert np.allclose(oracle.fft(oracle.fft(z), inverse=True), z, rtol=0, atol=1e-12)


@pytest.mark.parametrize("n", [8, 64, 1024, 4096])
def test_fft_fixed(oracle, n):
    d = load("fft.npz")
    q = d[f"fftx_in_{n}"]
    assert same(oracle.fft_fixed(q), d[f"fftx_fwd_{n}"])
    assert same(oracle.fft_fixed(q, inverse=True), d[f"fftx_inv_{n}"])


def test_fft_fixed_known_answers(oracle):
    d = load("fft.npz")
    ramp = np.zeros(16, dtype=np.int32)
    ramp[0::2] = 1000 * np.arange(8)
    f = oracle.fft_fixed(ramp)
    assert same(f, d["fftx_ramp_fwd"])
    assert list(f[:4]) == [28000, 0, -4001, 9655] and list(f[6:10]) == [-4000, 1657, -4000, 0]
    rt = oracle.fft_fixed(f, inverse=True)
    assert same(rt, d["fftx_ramp_roundtrip"])
    assert list(rt[0::2]) == [0, 1000, 2000, 3000, 4000, 5000, 5999, 6999]
    f = oracle.fft_fixed(d["fftx_sin_in"])
    assert same(f, d["fftx_sin_fwd"])
    assert (f[0], f[1], f[98], f[99]) == (10417, 0, -1660804, -4743492)
    rt = oracle.fft_fixed(f, inverse=True)
    assert same(rt, d["fftx_sin_roundtrip"]) and rt[20] == 1411


def test_synth_generator_properties(oracle):
    x = oracle.synth_f32(4, 4096, seed=0x11c0ffee)
    assert x.min() >= -1.0 and x.max() < 1.0 and abs(float(x.mean())) < 0.02
    # any slice regenerates identically (counter based): channel offset view
    y = oracle.synth_f32(2, 4096, seed=0x11c0ffee, chan0=2)
    assert np.array_equal(x[2:], y)
    s = oracle.synth_i16(2, 4096, seed=3)
    assert s.min() >= -16384 and s.max() <= 16383


def test_correlation(oracle):
    d = load("corr.npz")
    for n, p in ((64, 10), (300, 16), (1024, 32), (2048, 2047)):
        x, y = d[f"x_{n}"], d[f"y_{n}"]
        assert same(oracle.autocorr(x, p), d[f"auto_{n}_{p}"])
        assert same(oracle.crosscorr(x, y, p), d[f"cross_{n}_{p}"])
        assert same(oracle.autocorr_fast(x, p), d[f"fast_{n}_{p}"])
        assert oracle.corr_cof(x, y) == d[f"cof_{n}"][0]
    # the FFT form is the reference's own definition (first n bins, doubled), NOT the direct autocorrelation
    x = d["x_64"]
    assert not np.allclose(oracle.autocorr_fast(x, 10), oracle.autocorr(x, 10), rtol=1e-3)


def test_stft_frames(oracle):
    """llz_analysis_fft / llz_synthesis_fft (llz_asmodel.c:180-310): spectra and overlap-add output, streamed frame by
    frame, identical to the compiled reference"""
    d = load("stft.npz")
    for hint, frame_len, win in ((0, 8, po.HAMMING), (0, 64, po.BLACKMAN), (0, 256, po.KAISER), (1, 8, po.KAISER),
                                 (1, 128, po.HAMMING), (1, 512, po.BLACKMAN)):
        key = f"{hint}_{frame_len}_{win}"
        re, im = oracle.stft_analysis(hint, frame_len, win, d["x_" + key])
        assert same(re, d["re_" + key]) and same(im, d["im_" + key])
        assert re.shape[1] == (frame_len << (2 if hint == 0 else 1)) // 2 + 1
        assert same(oracle.stft_synthesis(hint, frame_len, win, re, im), d["syn_" + key])


def test_mdct(oracle):
    """llz_mdct.c: both windows, the three algorithms forward and inverse, and the TDAC frames of llz_asmodel.c:313-463,
    identical to the compiled reference"""
    d = load("mdct.npz")
    for n in (16, 64, 256, 2048):
        assert same(oracle.mdct_window(0, n), d[f"sine_{n}"]) and same(oracle.mdct_window(1, n), d[f"kbd_{n}"])
        x = d[f"x_{n}"]
        for t in (0, 1, 2):
            if t == 0 and n > 256:
                continue
            X = oracle.mdct(t, x)
            assert same(X, d[f"mdct{t}_{n}"]) and same(oracle.imdct(t, X), d[f"imdct{t}_{n}"]), (n, t)
        xi = d[f"xi_{n}"]
        for t in (0, 1, 2):
            if t == 0 and n > 256:
                continue
            Xi = oracle.mdct_fixed(t, xi)
            assert np.array_equal(Xi, d[f"mdctx{t}_{n}"]), (n, t)
            assert np.array_equal(oracle.mdct_fixed(t, Xi, inverse=True), d[f"imdctx{t}_{n}"]), (n, t)
    for frame_len, win in ((8, 0), (64, 1), (512, 0)):
        x = d[f"fx_{frame_len}_{win}"]
        X, y = oracle.mdct_frames(frame_len, win, x)
        assert same(X, d[f"fX_{frame_len}_{win}"]) and same(y, d[f"fy_{frame_len}_{win}"])
        # time-domain alias cancellation: the input comes back one frame late
        assert np.abs(y[frame_len:] - x[:-frame_len]).max() < 1e-13
