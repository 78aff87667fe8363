"""GPU parity: every kernel through the C ABI (include/*.h) against the CPU oracle and the golden fixtures.

Bit-exact for double (exact-order kernels), int16 and int32; float32 batch paths within 1e-5 RMS absolute and
relative to rms(reference) -- the tolerance BASELINE.json's north_star states.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from llzlab_amd import capi, filters  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-5   # RMS, north_star


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    assert capi.lib().llz_hip_device_count() >= 1, capi.last_error()
    torch.cuda.set_device(0)
    capi.check(capi.lib().llz_hip_set_device(0), "set_device")
    return torch.device("cuda:0")


def rms_check(got, ref, what):
    got = np.asarray(got, dtype=np.float64)
    err = float(np.sqrt(np.mean((got - ref) ** 2)))
    rel = err / max(float(np.sqrt(np.mean(ref ** 2))), 1e-30)
    assert err <= TOL and rel <= TOL, f"{what}: rms {err:.3g} rel {rel:.3g}"
    return err, rel


# ------------------------------------------------------------------------------------------------ design + synth
def test_host_design_matches_golden(dev):
    d = load("design.npz")
    kinds = ["lpf", "hpf", "bandpass", "bandstop"]
    for k, kind in enumerate(kinds):
        for win in range(3):
            for n in (15, 16, 63, 64, 257):
                got = filters.fir_design(kind, n, 0.2 if k >= 2 else 0.25, 0.4, win)
                assert np.array_equal(got, d[f"taps_k{k}_w{win}_n{n}"]), (kind, win, n)
    for win in range(3):
        assert np.array_equal(filters.window(win, 33), d[f"win_w{win}_n33"])


def test_synth_matches_oracle(dev, oracle):
    x = torch.empty(5, 3001, dtype=torch.float32, device=dev)
    filters.synth_f32(x, seed=0x11c0ffee, chan0=3)
    assert np.array_equal(x.cpu().numpy(), oracle.synth_f32(5, 3001, 0x11c0ffee, chan0=3))
    s = torch.empty(3, 2048, dtype=torch.int16, device=dev)
    filters.synth_i16(s, seed=9)
    assert np.array_equal(s.cpu().numpy(), oracle.synth_i16(3, 2048, 9))


# ------------------------------------------------------------------------------------------------ FIR
def test_fir_single_channel_exact_vs_golden(dev):
    d = load("fir_stream.npz")
    kinds = ["lpf", "hpf", "bandpass", "bandstop"]
    for k, kind in enumerate(kinds):
        for win in range(3):
            f = filters.FirFilter(kind, 64, 31, 0.2 if k >= 2 else 0.3, 0.45, win)
            y = np.concatenate([f.filter(d["x"][o:o + 64]) for o in range(0, len(d["x"]), 64)])
            tail = f.flush()
            f.close()
            assert np.array_equal(y, d[f"y_k{k}_w{win}"]), (kind, win)
            assert np.array_equal(tail, d[f"tail_k{k}_w{win}"]), (kind, win)
    f = filters.FirFilter("lpf", 512, 257, 0.1, 0.0, po.KAISER)
    x = d["x32_257"].astype(np.float64)
    y = np.concatenate([f.filter(x[o:o + 512]) for o in range(0, len(x), 512)])
    assert np.array_equal(y, d["y32_257"]) and np.array_equal(f.flush(), d["tail32_257"])
    f.close()


def test_fir_single_channel_rejects_other_frame_len(dev):
    f = filters.FirFilter("lpf", 64, 31, 0.3)
    with pytest.raises(capi.LlzError):
        f.filter(np.zeros(32))
    f.close()


@pytest.mark.parametrize("flt_len,algo", [(63, filters.FIR_ALGO_TIME), (257, filters.FIR_ALGO_TIME),
                                          (257, filters.FIR_ALGO_OVERLAP_SAVE), (200, filters.FIR_ALGO_OVERLAP_SAVE),
                                          (5, filters.FIR_ALGO_OVERLAP_SAVE), (1, filters.FIR_ALGO_TIME),
                                          (63, filters.FIR_ALGO_TIME_MFMA), (257, filters.FIR_ALGO_TIME_MFMA),
                                          (32, filters.FIR_ALGO_TIME_MFMA), (1, filters.FIR_ALGO_TIME_MFMA),
                                          (2, filters.FIR_ALGO_TIME_MFMA), (601, filters.FIR_ALGO_TIME_MFMA)])
@pytest.mark.parametrize("channels,n", [(3, 4096), (5, 1536 * 3), (2, 1000), (7, 2049), (2, 20001)])
def test_fir_mc_vs_oracle(dev, oracle, flt_len, algo, channels, n):
    taps = oracle.fir_design(po.LPF, flt_len, 0.2, 0.0, po.KAISER) if flt_len > 1 else np.array([0.75])
    x = torch.empty(channels, n, dtype=torch.float32, device=dev)
    filters.synth_f32(x, seed=flt_len + n)
    y = torch.empty_like(x)
    f = filters.FirFilterMC(channels, n, taps, algo=algo)
    assert f.algo == algo
    f.filter(x, y)
    ref = oracle.fir_batch_f32(x.cpu().numpy(), taps.astype(np.float32).astype(np.float64))
    rms_check(y.cpu().numpy(), ref, f"fir mc T={flt_len} algo={algo}")
    f.close()


@pytest.mark.parametrize("algo", [filters.FIR_ALGO_TIME, filters.FIR_ALGO_OVERLAP_SAVE, filters.FIR_ALGO_TIME_MFMA])
def test_fir_mc_streaming_frames_and_flush(dev, oracle, algo):
    """three equal frames == one long frame (history carried on the device), then flush == the filter's tail"""
    d = load("fir_stream.npz")
    taps = oracle.fir_design(po.LPF, 257, 0.1, 0.0, po.KAISER)
    x = np.stack([d["x32_257"], d["x32_257"][::-1].copy()])            # 2 channels x 1536
    f = filters.FirFilterMC(2, 512, taps, algo=algo)
    outs = []
    for o in range(0, 1536, 512):
        xi = torch.from_numpy(np.ascontiguousarray(x[:, o:o + 512])).to(dev)
        yi = torch.empty_like(xi)
        f.filter(xi, yi)
        outs.append(yi.cpu().numpy())
    y = np.concatenate(outs, axis=1)
    tail = torch.empty(2, 256, dtype=torch.float32, device=dev)
    f.flush(tail)
    f.close()
    rms_check(y[0], d["y32_257"], "streamed frames vs reference fixture")
    t32 = taps.astype(np.float32).astype(np.float64)
    full = oracle.fir_batch_f32(np.concatenate([x, np.zeros((2, 256), np.float32)], axis=1), t32)
    rms_check(y, full[:, :1536], "streamed frames vs oracle")
    assert np.sqrt(np.mean((tail.cpu().numpy() - full[:, 1536:]) ** 2)) <= TOL


def test_fir_mc_host_pointers_and_errors(dev, oracle):
    taps = oracle.fir_design(po.LPF, 63, 0.25, 0.0, po.HAMMING)
    x = oracle.synth_f32(4, 2048, 5)
    y = np.zeros_like(x)
    f = filters.FirFilterMC(4, 2048, taps, algo=filters.FIR_ALGO_TIME)
    assert f.algo == filters.FIR_ALGO_TIME
    f.filter(x, y)                                                     # numpy = host memory, staged by the library
    rms_check(y, oracle.fir_batch_f32(x, taps.astype(np.float32).astype(np.float64)), "host staging")
    with pytest.raises(capi.LlzError):
        capi.check(capi.lib().llz_fir_filter_mc(f.handle, x.ctypes.data, y.ctypes.data, 1024), "short frame")
    f.close()
    with pytest.raises(capi.LlzError):
        filters.FirFilterMC(4, 2048, np.ones(300), algo=filters.FIR_ALGO_OVERLAP_SAVE)
    assert filters.FirFilterMC(2, 64, np.ones(257)).algo == filters.FIR_ALGO_OVERLAP_SAVE
    assert filters.FirFilterMC(2, 64, np.ones(63)).algo == filters.FIR_ALGO_OVERLAP_SAVE      # AUTO: 33..257 taps
    assert filters.FirFilterMC(2, 64, np.ones(32)).algo == filters.FIR_ALGO_TIME
    assert filters.FirFilterMC(2, 64, np.ones(300)).algo == filters.FIR_ALGO_OVERLAP_SAVE_2048   # 258..513
    assert filters.FirFilterMC(2, 64, np.ones(1000)).algo == filters.FIR_ALGO_OVERLAP_SAVE_4096  # 514..1025
    assert filters.FirFilterMC(2, 64, np.ones(3000)).algo == filters.FIR_ALGO_OVERLAP_SAVE_8192  # 1026..6145
    assert filters.FirFilterMC(2, 64, np.ones(6145)).algo == filters.FIR_ALGO_OVERLAP_SAVE_8192
    assert filters.FirFilterMC(2, 64, np.ones(6146)).algo in (filters.FIR_ALGO_TIME_MFMA, filters.FIR_ALGO_TIME)   # time domain beyond
    assert filters.FirFilterMC(2, 64, np.ones(300), algo=filters.FIR_ALGO_TIME_MFMA).algo == filters.FIR_ALGO_TIME_MFMA


def test_fir_linearity_and_impulse_large(dev):
    """size-independent properties at a larger size: impulse -> taps, and filter(a x1 + x2) = a f(x1) + f(x2)"""
    taps = filters.fir_design("lpf", 257, 0.1, 0.0, po.KAISER)
    ch, n = 64, 1 << 16
    f = filters.FirFilterMC(ch, n, taps)
    x = torch.zeros(ch, n, dtype=torch.float32, device=dev)
    x[:, 1000] = 1.0
    y = torch.empty_like(x)
    f.filter(x, y)
    got = y[:, 1000:1257].cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got - taps[None, :])) < 2e-7
    assert float(y[:, :1000].abs().max()) < 1e-6 and float(y[:, 1257:].abs().max()) < 1e-6
    f.close()
    x1 = torch.empty(ch, n, dtype=torch.float32, device=dev)
    x2 = torch.empty_like(x1)
    filters.synth_f32(x1, seed=1)
    filters.synth_f32(x2, seed=2)
    ys = []
    for xi in (x1, x2, 0.5 * x1 + x2):
        f = filters.FirFilterMC(ch, n, taps)
        yi = torch.empty_like(xi)
        f.filter(xi, yi)
        ys.append(yi)
        f.close()
    assert float((ys[2] - (0.5 * ys[0] + ys[1])).pow(2).mean().sqrt()) < 1e-6


def test_fir_long_filter_full_size_impulse_and_linearity(dev):
    """the 3073-tap filter on the headline batch (4096 ch x 2^20: every pair of waves walks many segments of 16 jobs through
    k_fir_ols8k_f32), by properties that need no oracle: an impulse per channel, at a position that moves with the channel,
    returns the taps there and nothing elsewhere; filter(a x1 + x2) = a filter(x1) + filter(x2)"""
    T = 3073
    taps = filters.fir_design("lpf", T, 0.1, 0.0, po.KAISER)
    ch, n = 4096, 1 << 20
    f = filters.FirFilterMC(ch, n, taps)
    assert f.algo == filters.FIR_ALGO_OVERLAP_SAVE_8192
    x = torch.zeros(ch, n, dtype=torch.float32, device=dev)
    pos = (torch.arange(ch, device=dev) * 251) % (n - T)                      # spread over jobs, rows and block halves
    x[torch.arange(ch, device=dev), pos] = 1.0
    y = torch.empty_like(x)
    f.filter(x, y)
    idx = pos[:, None] + torch.arange(T, device=dev)[None, :]
    got = torch.gather(y, 1, idx)
    t32 = torch.from_numpy(taps.astype(np.float32)).to(dev)
    assert float((got - t32[None, :]).abs().max()) < 3e-7
    y.scatter_(1, idx, 0.0)                                                   # what is left is everything outside the responses
    assert float(y.abs().max()) < 1e-6
    del idx, got
    x2 = torch.empty_like(x)
    filters.synth_f32(x, seed=1)
    filters.synth_f32(x2, seed=2)
    y2 = torch.empty_like(x)
    f.close()
    outs = []
    for xi in (x, x2):
        g = filters.FirFilterMC(ch, n, taps)
        g.filter(xi, y if xi is x else y2)
        g.close()
    x.mul_(0.5).add_(x2)
    g = filters.FirFilterMC(ch, n, taps)
    g.filter(x, x2)                                                           # x2 <- filter(0.5 x1 + x2)
    g.close()
    y.mul_(0.5).add_(y2)
    assert float((x2 - y).pow(2).mean().sqrt()) < 1e-6


# ------------------------------------------------------------------------------------------------ IIR
def test_iir_single_channel_exact_vs_golden(dev):
    d = load("iir.npz")
    for a, b, yk, tk, fl in (("a2", "b2", "y2", "tail2", 100), ("a3", "b3", "y3", "tail3", 75),
                             ("a5", "b5", "y5", "tail5", 300)):
        f = filters.IirFilter(d[a], d[b])
        y = np.concatenate([f.filter(d["x"][o:o + fl]) for o in range(0, 300, fl)])
        assert np.array_equal(y, d[yk]), yk
        assert np.array_equal(f.flush(), d[tk]), tk
        f.close()
    f = filters.IirFilter(d["aq"], d["bq"])
    assert np.array_equal(f.filter(d["x32"].astype(np.float64)), d["yq"])
    f.close()


@pytest.mark.parametrize("stages", [1, 3, 8])
@pytest.mark.parametrize("channels,n", [(5, 1000), (70, 777), (64, 4096)])
def test_iir_cascade_mc_vs_oracle(dev, oracle, stages, channels, n):
    d = load("iir.npz")
    rows = [np.concatenate([d["b2"], d["a2"]])] * stages
    if stages >= 3:
        rows[1] = np.concatenate([d["bq"], d["aq"]])                   # one high-Q section (pole radius 0.99)
    coef = np.stack(rows)
    x = torch.empty(channels, n, dtype=torch.float32, device=dev)
    filters.synth_f32(x, seed=stages * 100 + channels)
    y = torch.empty_like(x)
    f = filters.IirCascadeMC(channels, coef)
    f.filter(x, y)
    ref = oracle.iir_cascade_batch_f32(x.cpu().numpy(), coef)
    rms_check(y.cpu().numpy(), ref, f"iir cascade S={stages}")
    f.close()


def test_iir_cascade_fixture_and_streaming(dev):
    d = load("iir.npz")
    coef = np.tile(np.concatenate([d["b2"], d["a2"]]), (8, 1))
    x = np.tile(d["x32"], (3, 1))
    f = filters.IirCascadeMC(3, coef)
    outs = []
    for o in (0, 500, 1100):                                           # ragged frame lengths: state carries
        e = {0: 500, 500: 1100, 1100: 2048}[o]
        xi = torch.from_numpy(np.ascontiguousarray(x[:, o:e])).to(dev)
        yi = torch.empty_like(xi)
        f.filter(xi, yi)
        outs.append(yi.cpu().numpy())
    f.close()
    y = np.concatenate(outs, axis=1)
    for c in range(3):
        rms_check(y[c], d["y_cascade8"], "cascade8 vs reference fixture")


# ------------------------------------------------------------------------------------------------ resample
def test_resample_single_channel_exact_vs_golden(dev):
    d = load("resample.npz")
    for (L, M) in [(1, 3), (2, 3), (3, 2), (147, 160), (160, 147)]:
        for win in range(3):
            seed, nin, frames = (int(v) for v in d[f"rs_{L}_{M}_w{win}_seed"])
            pcm = np.random.default_rng(seed).integers(-16384, 16384, nin * frames).astype(np.int16)
            r = filters.Resample(L, M, 1.0, win)
            assert r.bytes_in == 2 * nin
            out = np.concatenate([r.process(pcm[f * nin:(f + 1) * nin]) for f in range(frames)])
            r.close()
            assert np.array_equal(out, d[f"rs_{L}_{M}_w{win}_out"]), (L, M, win)
    r = filters.Resample(2, 3, 2.0, po.BLACKMAN)
    nin = r.bytes_in // 2
    out = np.concatenate([r.process(d["rs_clip_in"][f * nin:(f + 1) * nin]) for f in range(2)])
    r.close()
    assert np.array_equal(out, d["rs_clip_out"]) and out.max() == 32767 and out.min() == -32768
    with pytest.raises(capi.LlzError):
        filters.Resample(17, 1)
    r = filters.Resample(1, 3)
    with pytest.raises(capi.LlzError):
        r.process(np.zeros(100, dtype=np.int16))                       # wrong frame size: reference asserts
    r.close()


def test_decimate_interp_exact_vs_golden(dev):
    d = load("resample.npz")
    for M in (2, 3, 5):
        seed, nin, frames = (int(v) for v in d[f"dec_{M}_seed"])
        pcm = np.random.default_rng(seed).integers(-16384, 16384, nin * frames).astype(np.int16)
        r = filters.Decimate(M)
        out = np.concatenate([r.process(pcm[f * nin:(f + 1) * nin]) for f in range(frames)])
        r.close()
        assert np.array_equal(out, d[f"dec_{M}_out"]), M
    for L in (2, 3):
        seed, nin, frames = (int(v) for v in d[f"int_{L}_seed"])
        pcm = np.random.default_rng(seed).integers(-16384, 16384, nin * frames).astype(np.int16)
        r = filters.Interp(L)
        out = np.concatenate([r.process(pcm[f * nin:(f + 1) * nin]) for f in range(frames)])
        r.close()
        assert np.array_equal(out, d[f"int_{L}_out"]), L


@pytest.mark.parametrize("L,M,win", [(1, 3, po.BLACKMAN), (2, 3, po.HAMMING), (3, 2, po.KAISER), (147, 160, po.BLACKMAN)])
def test_resample_mc_i16_bit_exact(dev, oracle, L, M, win):
    info = oracle.rs_info(2, L, M, 1.0, win)
    nin = info["bytes_in"] // 2
    frames = 3 if nin < 4000 else 1
    ch = 5
    x = oracle.synth_i16(ch, nin * frames, seed=L * 31 + M)
    ref = oracle.rs_batch_i16(x, L, M, 1.0, win)
    r = filters.ResampleMC(ch, L, M, 1.0, win, filters.PCM_I16)
    assert r.Q == info["cols"] and np.array_equal(r.matrix(), info["matrix"])
    xd = torch.from_numpy(x).to(dev)
    yd = torch.empty(ch, ref.shape[1], dtype=torch.int16, device=dev)
    assert r.process(xd, yd) == ref.shape[1]
    assert np.array_equal(yd.cpu().numpy(), ref)
    r.close()
    # streaming in two calls gives the same samples
    if frames == 3:
        r = filters.ResampleMC(ch, L, M, 1.0, win, filters.PCM_I16)
        outs = []
        for (o, e) in ((0, nin), (nin, 3 * nin)):
            xi = torch.from_numpy(np.ascontiguousarray(x[:, o:e])).to(dev)
            yi = torch.empty(ch, (e - o) * L // M, dtype=torch.int16, device=dev)
            r.process(xi, yi)
            outs.append(yi.cpu().numpy())
        r.close()
        assert np.array_equal(np.concatenate(outs, axis=1), ref)


@pytest.mark.parametrize("M,win,gain", [(3, po.BLACKMAN, 1.0), (2, po.HAMMING, 1.0), (5, po.KAISER, 1.0), (3, po.BLACKMAN, 2.5)])
def test_resample_mc_i16_fast_within_one_lsb(dev, oracle, M, win, gain):
    """LLZ_PCM_I16_FAST: int16 in/out, the reference's result within one LSB.  Contract: every sample within
    one LSB of the reference's double-accumulate result, RMS deviation <= 1e-5 of full scale; streamed in two calls; gain 2.5
    drives the clamp at +-32767/-32768.  Since round 3 a FAST handle takes the bit-exact screened kernel where the screen
    accepts the taps (it is the faster one) -- then every sample is equal -- and the float32-sum matrix-core kernel otherwise
    (forced here with rs_i16_path = 1)"""
    info = oracle.rs_info(2, 1, M, gain, win)
    nin = info["bytes_in"] // 2 * 40
    ch = 7
    x = oracle.synth_i16(ch, nin, seed=M * 17)
    if gain > 1.0:
        x = (x.astype(np.int32) * 2).clip(-32768, 32767).astype(np.int16)          # full scale: the output clips
    ref = oracle.rs_batch_i16(x, 1, M, gain, win)
    for tuned in ({}, {"rs_i16_path": 1}):
        with capi.tuned(**tuned):
            r = filters.ResampleMC(ch, 1, M, gain, win, filters.PCM_I16_FAST)
            outs = []
            cut = (nin // 2) // M * M
            for (o, e) in ((0, cut), (cut, nin)):
                xi = torch.from_numpy(np.ascontiguousarray(x[:, o:e])).to(dev)
                yi = torch.empty(ch, (e - o) // M, dtype=torch.int16, device=dev)
                r.process(xi, yi)
                outs.append(yi.cpu().numpy())
            r.close()
        got = np.concatenate(outs, axis=1).astype(np.int32)
        diff = got - ref.astype(np.int32)
        assert np.abs(diff).max() <= 1, np.abs(diff).max()
        assert np.sqrt(np.mean(diff.astype(np.float64) ** 2)) <= 1e-5 * 32768
        assert np.mean(diff != 0) < 0.05                                    # a few percent of the samples sit on an edge
        if tuned:
            assert (diff != 0).any(), "the float32-sum kernel is expected to differ on some truncation edges"
    if gain > 1.0:
        assert (ref == 32767).any() or (ref == -32768).any()
    with pytest.raises(capi.LlzError):
        filters.ResampleMC(2, 2, 3, 1.0, win, filters.PCM_I16_FAST)    # decimators only


@pytest.mark.parametrize("L,M,win", [(1, 3, po.BLACKMAN), (2, 3, po.HAMMING), (3, 2, po.KAISER), (160, 147, po.BLACKMAN)])
def test_resample_mc_f32_vs_oracle(dev, oracle, L, M, win):
    ch = 4
    n_in = 3 * M * 512
    x = oracle.synth_f32(ch, n_in, seed=77)
    r = filters.ResampleMC(ch, L, M, 1.0, win, filters.PCM_F32)
    n_out = r.out_len(n_in)
    xd = torch.from_numpy(x).to(dev)
    yd = torch.empty(ch, n_out, dtype=torch.float32, device=dev)
    r.process(xd, yd)
    r.close()
    # oracle: same indexing with the float-rounded tap matrix the device uses, double accumulate
    ref = oracle.rs_batch_f32(x, L, M, 1.0, win)
    got = yd.cpu().numpy().astype(np.float64)
    err = float(np.sqrt(np.mean((got - ref) ** 2)))
    assert err <= TOL and err / float(np.sqrt(np.mean(ref ** 2))) <= TOL, err


@pytest.mark.parametrize("L,M", [(2, 3), (3, 2), (3, 4), (4, 3), (2, 1), (3, 1), (4, 1), (5, 3)])
def test_resample_mc_f32_small_ratios_streaming(dev, oracle, L, M):
    """the register-window kernel of the small ratios (k_resample_f32_win; 5:3 stays on the general kernel), streamed in
    calls of uneven length (history and the handle's stream position in use, tiles that start inside a call, several
    tiles per channel); against the oracle on the whole signal"""
    ch = 3
    lens = [1000 * M, 37 * M, 9001 * M, 5 * M]                          # a call takes whole periods of M inputs
    x = oracle.synth_f32(ch, sum(lens), seed=L * 10 + M)
    r = filters.ResampleMC(ch, L, M, 1.0, po.BLACKMAN, filters.PCM_F32)
    outs, o = [], 0
    for n_in in lens:
        n_out = r.out_len(n_in)
        assert n_out > 0
        xd = torch.from_numpy(np.ascontiguousarray(x[:, o:o + n_in])).to(dev)
        yd = torch.empty(ch, n_out, dtype=torch.float32, device=dev)
        r.process(xd, yd)
        outs.append(yd.cpu().numpy())
        o += n_in
    r.close()
    got = np.concatenate(outs, axis=1).astype(np.float64)
    ref = oracle.rs_batch_f32(x, L, M, 1.0, po.BLACKMAN)[:, :got.shape[1]]
    assert got.shape[1] >= sum(lens) * L // M - 1
    err = float(np.sqrt(np.mean((got - ref) ** 2)))
    assert err <= TOL and err / float(np.sqrt(np.mean(ref ** 2))) <= TOL, (L, M, err)


@pytest.mark.parametrize("form", ["phase-tile waves", "period-tile waves"])
@pytest.mark.parametrize("L,M", [(147, 160), (160, 147), (441, 320), (320, 441), (20, 147)])
def test_resample_mc_f32_matrix_core_streaming(dev, oracle, L, M, form):
    """large L and M on the fp32 matrix cores (resample_mfma.hip): a wave per phase tile walking spans of periods (441 and 320
    phases: more phase tiles than waves, taps reloaded per tile) and the first form with a wave per period tile; streamed in
    calls of uneven length -- one period, a few spans, a length that ends inside a span -- so that the history in front of a
    call, the walk's first and last span and the ragged tail are all in use; against the oracle on the whole signal"""
    ch = 3
    lens = [200 * M, M, 1037 * M, 49 * M]                               # a call takes whole periods of M inputs
    x = oracle.synth_f32(ch, sum(lens), seed=L + M)
    with capi.tuned(rs_mfma_form=1 if form == "period-tile waves" else -1):
        r = filters.ResampleMC(ch, L, M, 1.0, po.BLACKMAN, filters.PCM_F32)
        outs, o = [], 0
        for n_in in lens:
            n_out = r.out_len(n_in)
            assert n_out == n_in // M * L
            xd = torch.from_numpy(np.ascontiguousarray(x[:, o:o + n_in])).to(dev)
            yd = torch.empty(ch, n_out, dtype=torch.float32, device=dev)
            r.process(xd, yd)
            outs.append(yd.cpu().numpy())
            o += n_in
        r.close()
    got = np.concatenate(outs, axis=1).astype(np.float64)
    ref = oracle.rs_batch_f32(x, L, M, 1.0, po.BLACKMAN)[:, :got.shape[1]]
    err = float(np.sqrt(np.mean((got - ref) ** 2)))
    assert err <= TOL and err / float(np.sqrt(np.mean(ref ** 2))) <= TOL, (L, M, form, err)


def test_resample_mc_f32_integer_input_matches_int16_reference(dev, oracle):
    """integer-valued float input: trunc(clamp(float path)) equals the bit-exact int16 path except within ~1e-2
    of an integer boundary (float accumulate), SURVEY.md H3"""
    ch, n_in = 3, 1536 * 4
    xi = oracle.synth_i16(ch, n_in, seed=4)
    ref = oracle.rs_batch_i16(xi, 1, 3, 1.0, po.BLACKMAN)
    r = filters.ResampleMC(ch, 1, 3, 1.0, po.BLACKMAN, filters.PCM_F32)
    yd = torch.empty(ch, n_in // 3, dtype=torch.float32, device=dev)
    r.process(torch.from_numpy(xi.astype(np.float32)).to(dev), yd)
    r.close()
    got = np.trunc(np.clip(yd.cpu().numpy(), -32768, 32767)).astype(np.int16)
    assert np.max(np.abs(got.astype(np.int32) - ref)) <= 1
    assert np.mean(got != ref) < 0.01


# ------------------------------------------------------------------------------------------------ FFT
@pytest.mark.parametrize("n", [8, 64, 1024, 4096])
def test_fft_double_exact_vs_golden(dev, n):
    d = load("fft.npz")
    z = d[f"fft_in_{n}"].astype(np.complex128)
    f = filters.Fft(n)
    assert np.array_equal(f.fft(z), d[f"fft_fwd_{n}"])
    assert np.array_equal(f.ifft(z), d[f"fft_inv_{n}"])
    f.close()


@pytest.mark.parametrize("n", [8, 64, 1024, 4096])
def test_fft_fixed_bit_exact_vs_golden(dev, n):
    d = load("fft.npz")
    q = d[f"fftx_in_{n}"]
    f = filters.FftFixed(n)
    assert np.array_equal(f.fft(q), d[f"fftx_fwd_{n}"])
    assert np.array_equal(f.ifft(q), d[f"fftx_inv_{n}"])
    f.close()


def test_fft_fixed_known_answers_and_batch(dev, oracle):
    d = load("fft.npz")
    f = filters.FftFixed(1024)
    fw = f.fft(d["fftx_sin_in"])
    assert np.array_equal(fw, d["fftx_sin_fwd"]) and np.array_equal(f.ifft(fw), d["fftx_sin_roundtrip"])
    rng = np.random.default_rng(11)
    batch = rng.integers(-10000, 10001, (37, 2048)).astype(np.int32)
    bd = torch.from_numpy(batch).to(dev)
    f.fft_batch(bd, 37)
    ref = np.stack([oracle.fft_fixed(row) for row in batch])
    assert np.array_equal(bd.cpu().numpy(), ref)
    f.ifft_batch(bd, 37)
    ref2 = np.stack([oracle.fft_fixed(row, inverse=True) for row in ref])
    assert np.array_equal(bd.cpu().numpy(), ref2)
    f.close()
    f = filters.FftFixed(8)
    ramp = np.zeros(16, dtype=np.int32)
    ramp[0::2] = 1000 * np.arange(8)
    assert np.array_equal(f.fft(ramp), d["fftx_ramp_fwd"])
    f.close()


@pytest.mark.parametrize("n", [8, 256, 1024, 4096])
def test_fft_batch_f32_vs_oracle(dev, oracle, n):
    rng = np.random.default_rng(n)
    count = 9
    z = (rng.uniform(-1, 1, (count, n)) + 1j * rng.uniform(-1, 1, (count, n))).astype(np.complex64)
    zd = torch.from_numpy(z.view(np.float32).copy()).to(dev)
    f = filters.FftBatch(n)
    f.fft(zd, count)
    got = zd.cpu().numpy().view(np.complex64)
    ref = np.stack([oracle.fft(row.astype(np.complex128)) for row in z])
    assert np.sqrt(np.mean(np.abs(got - ref) ** 2)) / np.sqrt(np.mean(np.abs(ref) ** 2)) < 1e-6
    f.ifft(zd, count)                                                   # round trip = identity
    back = zd.cpu().numpy().view(np.complex64)
    assert np.sqrt(np.mean(np.abs(back - z) ** 2)) < 1e-6
    f.close()


@pytest.mark.parametrize("n,count", [(2, 5), (4, 3), (16, 130), (32, 65), (64, 100), (128, 17), (128, 333), (256, 47), (512, 5), (512, 70),
                                     (1024, 21), (2048, 3), (2048, 19), (4096, 2)])
def test_fft_batch_all_sizes_fixed_and_float(dev, oracle, n, count):
    """every power of two, batch counts that do not fill the last workgroup: Q15 bit-exact, float within tolerance"""
    rng = np.random.default_rng(n * 1000 + count)
    q = rng.integers(-8000, 8001, (count, 2 * n)).astype(np.int32)
    f = filters.FftFixed(n)
    qd = torch.from_numpy(q).to(dev)
    f.fft_batch(qd, count)
    ref = np.stack([oracle.fft_fixed(row) for row in q])
    assert np.array_equal(qd.cpu().numpy(), ref)
    f.ifft_batch(qd, count)
    assert np.array_equal(qd.cpu().numpy(), np.stack([oracle.fft_fixed(row, inverse=True) for row in ref]))
    f.close()
    if n >= 8:
        z = (rng.uniform(-1, 1, (count, n)) + 1j * rng.uniform(-1, 1, (count, n))).astype(np.complex64)
        zd = torch.from_numpy(z.view(np.float32).copy()).to(dev)
        fb = filters.FftBatch(n)
        fb.fft(zd, count)
        got = zd.cpu().numpy().view(np.complex64)
        refz = np.stack([oracle.fft(row.astype(np.complex128)) for row in z])
        assert np.sqrt(np.mean(np.abs(got - refz) ** 2)) / np.sqrt(np.mean(np.abs(refz) ** 2)) < 1e-6
        fb.ifft(zd, count)
        assert np.sqrt(np.mean(np.abs(zd.cpu().numpy().view(np.complex64) - z) ** 2)) < 1e-6
        fb.close()


@pytest.mark.parametrize("n", [2, 4, 16, 32, 128, 512, 2048])
def test_fft_double_exact_vs_oracle_other_sizes(dev, oracle, n):
    rng = np.random.default_rng(n)
    z = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    f = filters.Fft(n)
    assert np.array_equal(f.fft(z), oracle.fft(z))
    assert np.array_equal(f.ifft(z), oracle.fft(z, inverse=True))
    f.close()


def test_fft_fixed_wraps_like_the_reference_arithmetic(dev, oracle):
    """large inputs overflow int32 in the forward transform; the oracle wraps explicitly and so does the kernel"""
    rng = np.random.default_rng(99)
    q = rng.integers(-(1 << 30), 1 << 30, (3, 2 * 256)).astype(np.int32)
    f = filters.FftFixed(256)
    qd = torch.from_numpy(q).to(dev)
    f.fft_batch(qd, 3)
    assert np.array_equal(qd.cpu().numpy(), np.stack([oracle.fft_fixed(row) for row in q]))
    f.close()


@pytest.mark.parametrize("radius,theta", [(0.44, 1.1), (0.7, 0.5), (0.8, 2.0), (0.9, 0.3)])
@pytest.mark.parametrize("channels,n", [(3, 1024 * 6), (2, 1024 * 200)])
def test_iir_cascade_low_q_float32_path(dev, oracle, radius, theta, channels, n):
    """cascades whose sections all have a small rounding-noise gain (measured at init) run the pipelined kernel in
    float32; the 0.9-radius set fails that check and stays in double.  Either way the result must meet the tolerance
    against the double oracle, in one call, split along time (2 channels x 200 chunks) and across calls"""
    a1, a2 = -2 * radius * np.cos(theta), radius ** 2
    g = (1 + a1 + a2) / 4                                               # unit DC gain with b = g * [1, 2, 1]
    coef = np.tile(np.array([g, 2 * g, g, 1.0, a1, a2]), (8, 1))
    x = oracle.synth_f32(channels, n + 2048, seed=int(radius * 100))
    ref = oracle.iir_cascade_batch_f32(x, coef)
    f = filters.IirCascadeMC(channels, coef)
    assert f.precision == (64 if radius >= 0.9 else 32)
    outs = []
    for (o, e) in ((0, n), (n, n + 2048)):
        xi = torch.from_numpy(np.ascontiguousarray(x[:, o:e])).to(dev)
        yi = torch.empty_like(xi)
        f.filter(xi, yi)
        outs.append(yi.cpu().numpy())
    f.close()
    got = np.concatenate(outs, axis=1).astype(np.float64)
    err, scale = float(np.sqrt(np.mean((got - ref) ** 2))), float(np.sqrt(np.mean(ref ** 2)))
    # (radius 0.8 at 2 rad resonates far above the DC gain: eight sections give an output RMS of ~60, so the bound is
    # relative to the signal there and absolute at unit scale)
    assert err <= TOL * max(1.0, scale) and err / scale <= TOL, (radius, theta, err, scale)


@pytest.mark.parametrize("stages,radius,theta,chunks", [(8, 0.44, 1.1, 64), (3, 0.7, 0.5, 64), (16, 0.5, 2.5, 64),
                                                        (8, 0.99, 0.3, 256), (2, 0.95, 1.0, 128)])
def test_iir_cascade_wave_form(dev, oracle, stages, radius, theta, chunks):
    """many channels x a cascade of up to 8 sections with a short memory: a wave owns a (channel, time segment) and runs all
    sections in registers (k_iir_cascade_wave).  Two calls (state hand-over), ragged tail through the per-channel kernel, spot
    channels against the double oracle"""
    a1, a2 = -2 * radius * np.cos(theta), radius ** 2
    g = (1 + a1 + a2) / 4
    coef = np.tile(np.array([g, 2 * g, g, 1.0, a1, a2]), (stages, 1))
    channels, n = 2048, 1024 * chunks + 100
    f = filters.IirCascadeMC(channels, coef)
    # both precisions have the wave form (16 sections: pipelined).  0.95 at 1 rad has a noise gain of 7.6: float32
    assert f.precision == (64 if radius == 0.99 else 32)
    sel = [0, 1, 1000, 2047]
    xs, ys = [], []
    for call in range(2):
        x = torch.empty(channels, n, dtype=torch.float32, device=dev)
        filters.synth_f32(x, seed=40 + call)
        y = torch.empty_like(x)
        f.filter(x, y)
        xs.append(x[sel].cpu().numpy())
        ys.append(y[sel].cpu().numpy())
    f.close()
    ref = oracle.iir_cascade_batch_f32(np.concatenate(xs, axis=1), coef)
    got = np.concatenate(ys, axis=1).astype(np.float64)
    err, scale = float(np.sqrt(np.mean((got - ref) ** 2))), float(np.sqrt(np.mean(ref ** 2)))
    assert err <= TOL * max(1.0, scale) and err / scale <= TOL, (err, scale)


@pytest.mark.parametrize("stages", [2, 4, 5, 6, 7, 8])
def test_iir_cascade_wave_distinct_sections(dev, oracle, stages):
    """every section different (radius, angle, zeros): the packed float32 kernel (even counts) fetches a section's tables one
    section ahead into alternating register sets, so a mix-up between sections must show here; odd counts take the
    unpacked kernel.  Two calls, segments along time (few channels) and the many-channel shape"""
    rows = []
    for k in range(stages):
        r, th = 0.30 + 0.06 * k, 0.4 + 0.33 * k
        a1, a2 = -2 * r * np.cos(th), r * r
        b = np.array([1.0, 0.3 - 0.2 * k, 0.1 * k]) * (1 + a1 + a2) / (1.3 - 0.1 * k)
        rows.append(np.concatenate([b, [1.0, a1, a2]]))
    coef = np.array(rows)
    for channels, n in ((2048, 1024 * 16 + 40), (24, 1024 * 300)):
        f = filters.IirCascadeMC(channels, coef)
        assert f.precision == 32
        sel = [0, channels // 2 + 1, channels - 1]
        xs, ys = [], []
        for call in range(2):
            x = torch.empty(channels, n, dtype=torch.float32, device=dev)
            filters.synth_f32(x, seed=70 + call + stages)
            y = torch.empty_like(x)
            f.filter(x, y)
            xs.append(x[sel].cpu().numpy())
            ys.append(y[sel].cpu().numpy())
        f.close()
        ref = oracle.iir_cascade_batch_f32(np.concatenate(xs, axis=1), coef)
        got = np.concatenate(ys, axis=1).astype(np.float64)
        err, scale = float(np.sqrt(np.mean((got - ref) ** 2))), float(np.sqrt(np.mean(ref ** 2)))
        assert err <= TOL * max(1.0, scale) and err / scale <= TOL, (stages, channels, err, scale)


@pytest.mark.parametrize("stages", [1, 3, 4, 7, 8])
def test_iir_cascade_wave_double_distinct_sections(dev, oracle, stages):
    """high-Q sections, every one different (radius, angle, zeros, gain): the double kernel with 32 samples per lane and the
    b0 gains folded out (k_iir_cascade_wave_pf64w) fetches coefficients a section ahead into alternating register sets and
    keeps its states scaled by the gains still to come; frames of whole 2048-sample chunks + one 1024-sample chunk + a ragged
    tail (three kernels share the state), two calls, many channels and few (segments along time)"""
    rows = []
    for k in range(stages):
        r, th = 0.99 - 0.004 * k, 0.25 + 0.31 * k
        a1, a2 = -2 * r * np.cos(th), r * r
        b = np.array([1.0, 0.3 - 0.2 * k, 0.1 * k]) * (1 + a1 + a2) * (0.7 + 0.2 * k)
        rows.append(np.concatenate([b, [1.0, a1, a2]]))
    coef = np.array(rows)
    for channels, n in ((2048, 2048 * 40 + 1024 + 40), (24, 2048 * 600 + 1024)):
        f = filters.IirCascadeMC(channels, coef)
        assert f.precision == 64
        sel = [0, channels // 2 + 1, channels - 1]
        xs, ys = [], []
        for call in range(2):
            x = torch.empty(channels, n, dtype=torch.float32, device=dev)
            filters.synth_f32(x, seed=170 + call + stages)
            y = torch.empty_like(x)
            f.filter(x, y)
            xs.append(x[sel].cpu().numpy())
            ys.append(y[sel].cpu().numpy())
        f.close()
        ref = oracle.iir_cascade_batch_f32(np.concatenate(xs, axis=1), coef)
        got = np.concatenate(ys, axis=1).astype(np.float64)
        err, scale = float(np.sqrt(np.mean((got - ref) ** 2))), float(np.sqrt(np.mean(ref ** 2)))
        assert err <= TOL * max(1.0, scale) and err / scale <= TOL, (stages, channels, err, scale)


def test_iir_cascade_few_channels_split_along_time(dev, oracle):
    """few channels and a long frame: the pipelined kernel splits each channel into time segments that start a measured
    warm-up early from the zero state (the cascade's memory, probed at init); the result must still match the sequential
    oracle, including the high-Q section, and the state handed to the next call must be the true one"""
    d = load("iir.npz")
    coef = np.stack([np.concatenate([d["bq"], d["aq"]]), np.concatenate([d["b2"], d["a2"]]),
                     np.concatenate([d["bq"], d["aq"]])])
    channels, n = 2, 1024 * 240
    x = oracle.synth_f32(channels, n + 4096, seed=5)
    ref = oracle.iir_cascade_batch_f32(x, coef)
    f = filters.IirCascadeMC(channels, coef)
    xd = torch.from_numpy(np.ascontiguousarray(x[:, :n])).to(dev)
    yd = torch.empty_like(xd)
    f.filter(xd, yd)
    x2 = torch.from_numpy(np.ascontiguousarray(x[:, n:])).to(dev)
    y2 = torch.empty_like(x2)
    f.filter(x2, y2)                                                    # continues from the state the segments left
    f.close()
    got = np.concatenate([yd.cpu().numpy(), y2.cpu().numpy()], axis=1)
    rms_check(got, ref, "iir split along time")
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("channels,n", [(3, 1024 * 5), (2, 1024 * 3 + 777), (130, 2048)])
def test_iir_cascade_pipelined_path_long(dev, oracle, channels, n):
    """whole 1024-sample chunks run through the stage-pipelined kernel, the remainder through the per-channel one;
    two calls in a row exercise the state hand-over between the two"""
    d = load("iir.npz")
    coef = np.stack([np.concatenate([d["b2"], d["a2"]]), np.concatenate([d["bq"], d["aq"]]),
                     np.concatenate([d["b2"], d["a2"]]), np.concatenate([d["bq"], d["aq"]]),
                     np.concatenate([d["b2"], d["a2"]])])
    x = oracle.synth_f32(channels, 2 * n, seed=n)
    ref = oracle.iir_cascade_batch_f32(x, coef)
    f = filters.IirCascadeMC(channels, coef)
    outs = []
    for o in (0, n):
        xi = torch.from_numpy(np.ascontiguousarray(x[:, o:o + n])).to(dev)
        yi = torch.empty_like(xi)
        f.filter(xi, yi)
        outs.append(yi.cpu().numpy())
    f.close()
    rms_check(np.concatenate(outs, axis=1), ref, "pipelined iir, two frames")


# ------------------------------------------------------------------------------------------------ correlation (8f rank 1)
def test_correlation_reference_symbols_exact(dev):
    d = load("corr.npz")
    for n, p in ((64, 10), (300, 16), (1024, 32), (2048, 2047)):
        x, y = d[f"x_{n}"], d[f"y_{n}"]
        assert np.array_equal(filters.autocorr(x, p), d[f"auto_{n}_{p}"])
        assert np.array_equal(filters.crosscorr(x, y, p), d[f"cross_{n}_{p}"])
        assert np.array_equal(filters.autocorr_fast(x, p), d[f"fast_{n}_{p}"])
        assert filters.corr_cof(x, y) == d[f"cof_{n}"][0]
    with pytest.raises(capi.LlzError):
        capi.check_handle(capi.lib().llz_autocorr_fast_init(4096), "too long")


@pytest.mark.parametrize("frames,n,p", [(7, 300, 16), (3, 1024, 32), (5, 20000, 40), (2, 64, 63), (130, 512, 255),
                                        # the register form for short lag ranges (k_autocorr_reg_f32<1..4>): every neighbour count,
                                        # lengths that are no multiple of 8, shorter than a chunk, exactly one chunk, p = 0
                                        (9, 1021, 0), (300, 1000, 7), (33, 496, 8), (5, 497, 9), (1000, 480, 16), (4, 3001, 17),
                                        (6, 61, 24), (40, 2048, 25), (3, 10, 9), (2, 4096, 32)])
def test_autocorr_mc_direct_vs_oracle(dev, oracle, frames, n, p):
    x = oracle.synth_f32(frames, n, seed=n + p)
    ref = np.stack([oracle.autocorr(row.astype(np.float64), p) for row in x[:64]])
    xd = torch.from_numpy(x).to(dev)
    rd = torch.empty(frames, p + 1, dtype=torch.float32, device=dev)
    filters.autocorr_mc(xd, rd, p)
    got = rd.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got[:64] - ref)) <= 1e-5 * ref[:, 0].max()    # relative to the zero-lag energy
    if p <= 32:
        # the LDS-window kernel (every p) must agree with the register form on every frame
        with capi.tuned(acf_lds=1):
            r1 = torch.empty_like(rd)
            filters.autocorr_mc(xd, r1, p)
        assert np.max(np.abs(r1.cpu().numpy().astype(np.float64) - got)) <= 2e-6 * max(1.0, np.abs(got).max())
    r2 = np.zeros((frames, p + 1), dtype=np.float32)
    filters.autocorr_mc(x, r2, p)                                       # host buffers
    assert np.array_equal(r2, rd.cpu().numpy())


@pytest.mark.parametrize("frames,n,p", [(9, 300, 16), (4, 1024, 32), (3, 2048, 100), (70000, 16, 8),
                                        # fft_len 2048 runs as two 1024-point complex transforms (k_acf2048_f32): even / odd
                                        # frame lengths, p below and above the pruned-inverse limit of 64, partial groups
                                        (5000, 1024, 16), (7, 1000, 63), (3, 513, 200), (6, 777, 64), (5, 1023, 1500),
                                        (11, 600, 0),
                                        # fft_len 128 and 512 on the E = 8 / 16 lane groups (k_acf_sq_f32)
                                        (50, 64, 20), (40, 200, 33), (300, 256, 16), (9, 130, 255), (5, 33, 3), (70, 255, 500),
                                        # fft_len 1024 (k_acf1024_f32): pruned (p < 32) and full inverse
                                        (100, 512, 16), (13, 300, 31), (6, 257, 32), (9, 480, 1000),
                                        # fft_len 4096 = one complex 2048-point transform on a whole wave (k_acf4096_f32): odd
                                        # and even lengths, lags into the second half of the inverse, a partial last workgroup
                                        (5, 1025, 16), (9, 2000, 63), (4, 1500, 64), (3, 2047, 3000), (130, 1800, 4095),
                                        (2, 2048, 2049), (6, 1026, 0)])
def test_autocorr_fast_mc_vs_oracle(dev, oracle, frames, n, p):
    if frames > 1000:
        x = np.tile(oracle.synth_f32(8, n, seed=3), (frames // 8, 1))
        frames = x.shape[0]
    else:
        x = oracle.synth_f32(frames, n, seed=n)
    uniq = x[:8] if frames > 1000 else x
    ref_u = np.stack([oracle.autocorr_fast(row.astype(np.float64), p) for row in uniq])
    f = filters.AutocorrFastMC(frames, n)
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    rd = torch.empty(frames, p + 1, dtype=torch.float32, device=dev)
    f.run(xd, rd, p)
    f.close()
    got = rd.cpu().numpy().astype(np.float64)
    ref = np.tile(ref_u, (frames // 8, 1)) if frames > 1000 else ref_u
    assert np.max(np.abs(got - ref)) <= 1e-5 * np.abs(ref).max()


# ------------------------------------------------------------------------------------------------ PCM ingest (8f rank 2)
@pytest.mark.parametrize("channels,n", [(1, 5000), (1, 4099), (2, 48000), (2, 4099), (3, 1001), (6, 1000), (64, 4097), (64, 4096),
                                        (100, 333), (4096, 70), (4096, 128)])
def test_pcm_deinterleave_and_back_exact(dev, oracle, channels, n):
    """against the oracle's PCM functions (orc_pcm_*: planar = interleaved / 32768, exact in float32; back: clamp, then
    the C conversion toward zero, reference llz_resample.c:596-601)"""
    rng = np.random.default_rng(channels * 7 + n)
    il = rng.integers(-32768, 32768, (n, channels)).astype(np.int16)
    ild = torch.from_numpy(il).to(dev)
    pl = torch.empty(channels, n, dtype=torch.float32, device=dev)
    filters.pcm_deinterleave(ild, pl)
    assert np.array_equal(pl.cpu().numpy(), oracle.pcm_deinterleave(il))
    back = torch.empty(n, channels, dtype=torch.int16, device=dev)
    filters.pcm_interleave(pl, back)
    assert np.array_equal(back.cpu().numpy(), il)                      # round trip is the identity
    # clamp + truncation on out-of-range and fractional values
    x = rng.uniform(-1.5, 1.5, (channels, n)).astype(np.float32)
    out = np.zeros((n, channels), dtype=np.int16)
    filters.pcm_interleave(x, out)                                      # host buffers
    assert np.array_equal(out, oracle.pcm_interleave(x))


# ------------------------------------------------------------------------------------------------ windowed-FFT frames (8f rank 3)
STFT_CASES = ((0, 8, po.HAMMING), (0, 64, po.BLACKMAN), (0, 256, po.KAISER), (1, 8, po.KAISER), (1, 128, po.HAMMING),
              (1, 512, po.BLACKMAN))


def test_stft_reference_symbols_exact(dev):
    """llz_analysis_fft / llz_synthesis_fft through the product library, frame by frame, against the fixtures generated
    from the compiled reference: bit-identical (framing on the host in the reference's statement order, transforms by the
    exact-order double kernel)"""
    d = load("stft.npz")
    for hint, frame_len, win in STFT_CASES:
        key = f"{hint}_{frame_len}_{win}"
        x = d["x_" + key]
        a = filters.AnalysisFft(hint, frame_len, win)
        s = filters.SynthesisFft(hint, frame_len, win)
        for f in range(len(x) // frame_len):
            re, im = a.frame(x[f * frame_len:(f + 1) * frame_len])
            assert np.array_equal(re, d["re_" + key][f]) and np.array_equal(im, d["im_" + key][f]), (key, f)
            xo = s.frame(re, im)
            assert np.array_equal(xo, d["syn_" + key][f * frame_len:(f + 1) * frame_len]), (key, f)
        a.close()
        s.close()
    with pytest.raises(capi.LlzError):
        filters.AnalysisFft(0, 48)                                      # fft_len 192 is not a power of two
    with pytest.raises(capi.LlzError):
        filters.SynthesisFft(7, 64)


@pytest.mark.parametrize("hint,frame_len,win", [(0, 2, po.HAMMING), (0, 64, po.BLACKMAN), (0, 512, po.KAISER),
                                                (1, 4, po.KAISER), (1, 128, po.HAMMING), (1, 1024, po.BLACKMAN),
                                                (0, 256, po.KAISER), (1, 512, po.HAMMING),       # fft_len 1024 twice
                                                (0, 128, po.HAMMING), (1, 256, po.KAISER)])      # fft_len 512 twice
@pytest.mark.parametrize("channels,frames", [(3, 7), (2, 70), (5, 1)])
def test_stft_mc_vs_oracle_streaming(dev, oracle, hint, frame_len, win, channels, frames):
    """batch analysis and synthesis against the oracle run channel by channel, in TWO calls so that the history and the
    overlap-add tail carried by the handle are exercised; the synthesis input is the oracle's own spectra"""
    rng = np.random.default_rng(hint * 1000 + frame_len + channels)
    n = frames * frame_len
    x = rng.uniform(-1, 1, (channels, 2 * n)).astype(np.float32)
    ref = [oracle.stft_analysis(hint, frame_len, win, row.astype(np.float64)) for row in x]
    ref_re = np.stack([r[0] for r in ref])
    ref_im = np.stack([r[1] for r in ref])
    ref_x = np.stack([oracle.stft_synthesis(hint, frame_len, win, ref_re[c].astype(np.float32), ref_im[c].astype(np.float32))
                      for c in range(channels)])
    f = filters.StftMC(channels, hint, frame_len, win)
    bins = f.bins
    assert bins == (frame_len << (2 if hint == 0 else 1)) // 2 + 1
    got_re, got_im, got_x = [], [], []
    for half in range(2):
        xd = torch.from_numpy(np.ascontiguousarray(x[:, half * n:(half + 1) * n])).to(dev)
        re = torch.empty(channels, frames, bins, dtype=torch.float32, device=dev)
        im = torch.empty_like(re)
        f.analysis(xd, re, im)
        got_re.append(re.cpu().numpy())
        got_im.append(im.cpu().numpy())
        sre = torch.from_numpy(np.ascontiguousarray(ref_re[:, half * frames:(half + 1) * frames]).astype(np.float32)).to(dev)
        sim = torch.from_numpy(np.ascontiguousarray(ref_im[:, half * frames:(half + 1) * frames]).astype(np.float32)).to(dev)
        xo = torch.empty(channels, n, dtype=torch.float32, device=dev)
        f.synthesis(sre, sim, xo)
        got_x.append(xo.cpu().numpy())
    got_re, got_im = np.concatenate(got_re, axis=1), np.concatenate(got_im, axis=1)
    got_x = np.concatenate(got_x, axis=1)
    scale = max(np.sqrt(np.mean(ref_re ** 2 + ref_im ** 2)), 1e-30)
    err = np.sqrt(np.mean((got_re - ref_re) ** 2 + (got_im - ref_im) ** 2))
    assert err <= 1e-5 * max(scale, 1.0) and err / scale <= 1e-5, (err, scale)
    # with one or two frames the overlap-add sum sits on the window's edge (output ~1e-9 of the input): measure the error
    # against the larger of the output and a -26 dB floor of the unit-scale input, not against a vanishing output
    err_x = float(np.sqrt(np.mean((got_x - ref_x) ** 2)))
    assert err_x <= TOL and err_x / max(float(np.sqrt(np.mean(ref_x ** 2))), 0.05) <= TOL, err_x
    f.close()


@pytest.mark.parametrize("hint,frame_len", [(0, 512), (1, 1024), (0, 128), (1, 256)])
def test_stft_synthesis_half_size_transform_agrees_with_full(dev, hint, frame_len):
    """fft_len 512 and 2048: the synthesis takes ONE half-size complex inverse transform per real frame (k_stft_synthesis_reg_f32<..,
    HALF>); the full-size form (llz_hip_tune("stft_full", 1)) must give the same samples to float32 rounding, on spectra that
    are NOT Hermitian-consistent at bins 0 and N/2 (their imaginary parts do not reach a real output in either form)"""
    channels, frames = 3, 37
    rng = np.random.default_rng(frame_len + hint)
    bins = (frame_len << (2 if hint == 0 else 1)) // 2 + 1
    re = rng.uniform(-1, 1, (channels, frames, bins)).astype(np.float32)
    im = rng.uniform(-1, 1, (channels, frames, bins)).astype(np.float32)
    outs = []
    for full in (-1, 1):
        capi.tune("stft_full", full)
        try:
            f = filters.StftMC(channels, hint, frame_len, po.BLACKMAN)
            xo = torch.empty(channels, frames * frame_len, dtype=torch.float32, device=dev)
            f.synthesis(torch.from_numpy(re).to(dev), torch.from_numpy(im).to(dev), xo)
            outs.append(xo.cpu().numpy())
            f.close()
        finally:
            capi.tune("stft_full", -1)
    assert np.abs(outs[0] - outs[1]).max() <= 2e-5 * max(1.0, np.abs(outs[1]).max())


def test_stft_mc_host_buffers_and_many_runs(dev, oracle):
    """host numpy buffers, and enough frames that one channel is split over several workgroup runs (warm-up frames and
    dropped blocks in the synthesis kernel)"""
    hint, frame_len, win, channels, frames = 0, 16, po.BLACKMAN, 2, 700
    rng = np.random.default_rng(77)
    x = rng.uniform(-1, 1, (channels, frames * frame_len)).astype(np.float32)
    f = filters.StftMC(channels, hint, frame_len, win)
    re = np.zeros((channels, frames, f.bins), dtype=np.float32)
    im = np.zeros_like(re)
    f.analysis(x, re, im)
    ref = [oracle.stft_analysis(hint, frame_len, win, row.astype(np.float64)) for row in x]
    assert np.abs(re - np.stack([r[0] for r in ref])).max() <= 2e-5
    assert np.abs(im - np.stack([r[1] for r in ref])).max() <= 2e-5
    xo = np.zeros_like(x)
    f.synthesis(re, im, xo)
    ref_x = np.stack([oracle.stft_synthesis(hint, frame_len, win, re[c], im[c]) for c in range(channels)])
    rms_check(xo, ref_x, "stft synthesis, 700 frames")
    f.close()


# ------------------------------------------------------------------------------------------------ MDCT (8f rank 4)
def test_mdct_reference_symbols_exact(dev):
    """llz_mdct / llz_imdct (three algorithms), the two windows and the TDAC frames through the product library against
    the fixtures generated from the compiled reference: bit-identical"""
    d = load("mdct.npz")
    for n in (16, 64, 256, 2048):
        assert np.array_equal(filters.mdct_window(capi.MDCT_SINE, n), d[f"sine_{n}"])
        assert np.array_equal(filters.mdct_window(capi.MDCT_KBD, n), d[f"kbd_{n}"])
        x = d[f"x_{n}"]
        for t in (capi.MDCT_ORIGIN, capi.MDCT_FFT, capi.MDCT_FFT4):
            if t == capi.MDCT_ORIGIN and n > 256:
                continue
            m = filters.Mdct(t, n)
            X = m.forward(x)
            assert np.array_equal(X, d[f"mdct{t}_{n}"]), (n, t)
            assert np.array_equal(m.inverse(X), d[f"imdct{t}_{n}"]), (n, t)
            m.close()
        xi = d[f"xi_{n}"]
        for t in (0, 1, 2):                                             # fixed point: int32 data, Q15 tables
            if t == 0 and n > 256:
                continue
            m = filters.MdctFixed(t, n)
            Xi = m.forward(xi)
            assert np.array_equal(Xi, d[f"mdctx{t}_{n}"]), (n, t)
            assert np.array_equal(m.inverse(Xi), d[f"imdctx{t}_{n}"]), (n, t)
            m.close()
    for frame_len, win in ((8, 0), (64, 1), (512, 0)):
        x = d[f"fx_{frame_len}_{win}"]
        a, s = filters.AnalysisMdct(frame_len, win), filters.SynthesisMdct(frame_len, win)
        for f in range(len(x) // frame_len):
            X = a.frame(x[f * frame_len:(f + 1) * frame_len])
            assert np.array_equal(X, d[f"fX_{frame_len}_{win}"][f])
            assert np.array_equal(s.frame(X), d[f"fy_{frame_len}_{win}"][f * frame_len:(f + 1) * frame_len])
        a.close()
        s.close()
    with pytest.raises(capi.LlzError):
        filters.Mdct(capi.MDCT_ORIGIN, 4096)                            # the O(N^2) type is capped at 2048
    with pytest.raises(capi.LlzError):
        filters.Mdct(5, 64)


@pytest.mark.parametrize("t,n,count", [(0, 16, 7), (0, 256, 40), (1, 16, 3), (1, 256, 129), (1, 4096, 5), (2, 8, 4), (2, 64, 1000),
                                       (2, 2048, 37), (2, 16384, 3), (2, 16, 1), (2, 32, 2049), (2, 512, 301), (2, 8192, 2),
                                       (1, 8, 5), (1, 64, 1000), (1, 2048, 33)])
def test_mdct_fixed_batch_vs_oracle(dev, oracle, t, n, count):
    """llz_mdct_fixed_batch / llz_imdct_fixed_batch: `count` frames per call on the device kernels (mdct_q15.hip + the Q15
    transform), device tensors in place and host arrays staged, against the oracle's frame-by-frame llz_mdct_fixed (itself
    pinned by the reference's fixtures): bit-exact, including inputs large enough to wrap the int32 sums"""
    rng = np.random.default_rng(1000 * t + n + count)
    x = rng.integers(-(1 << 20), 1 << 20, (count, n), dtype=np.int32)
    x[0, : n // 2] = rng.integers(-(1 << 31), (1 << 31) - 1, n // 2, dtype=np.int64).astype(np.int32)   # wrap-around case
    uniq = sorted(set([0, 1, count // 2, count - 1]) & set(range(count)))
    m = filters.MdctFixed(t, n)
    xd = torch.from_numpy(x).to(dev)
    Xd = torch.empty(count, n // 2, dtype=torch.int32, device=dev)
    m.forward_batch(xd, Xd)
    yd = torch.empty(count, n, dtype=torch.int32, device=dev)
    m.inverse_batch(Xd, yd)
    torch.cuda.synchronize()
    X, y = Xd.cpu().numpy(), yd.cpu().numpy()
    if t in (1, 2):
        # the two FFT forms are one launch each (k_mdct1_q15 / k_mdct4_q15); as three launches they must give the same integers on
        # every frame
        with capi.tuned(mdctq_steps=1):
            X3, y3 = torch.empty_like(Xd), torch.empty_like(yd)
            m.forward_batch(xd, X3)
            m.inverse_batch(Xd, y3)
        assert torch.equal(X3, Xd) and torch.equal(y3, yd)
    for c in uniq:
        ref_X = oracle.mdct_fixed(t, x[c])
        assert np.array_equal(X[c], ref_X), (t, n, c)
        assert np.array_equal(y[c], oracle.mdct_fixed(t, ref_X, inverse=True)), (t, n, c)
    # host arrays go through the staging buffers; one frame per call is the reference's own symbol
    Xh = np.zeros((count, n // 2), dtype=np.int32)
    m.forward_batch(x, Xh)
    assert np.array_equal(Xh, X)
    assert np.array_equal(m.forward(x[count - 1]), X[count - 1])
    assert np.array_equal(m.inverse(X[0]), y[0])
    assert capi.lib().llz_mdct_fixed_batch(m.handle, xd.data_ptr(), Xd.data_ptr(), 0) < 0          # no frames: refused
    assert capi.lib().llz_mdct_fixed_batch(m.handle, xd.data_ptr(), xd.data_ptr(), 1) < 0          # in place: refused
    m.close()


@pytest.mark.parametrize("n,count", [(32, 5), (64, 1000), (256, 33), (2048, 9), (8192, 3),
                                     # all six register-transform sizes (k_mdct_reg_f32), counts that leave partial workgroups
                                     (256, 1000), (512, 77), (1024, 50), (2048, 130), (4096, 21), (8192, 19)])
def test_mdct_batch_vs_oracle(dev, oracle, n, count):
    rng = np.random.default_rng(n + count)
    x = rng.uniform(-1, 1, (count, n)).astype(np.float32)
    uniq = min(count, 6)
    ref = np.stack([oracle.mdct(2, row.astype(np.float64)) for row in x[:uniq]])
    m = filters.MdctBatch(n)
    xd = torch.from_numpy(x).to(dev)
    Xd = torch.empty(count, n // 2, dtype=torch.float32, device=dev)
    m.forward(xd, Xd)
    got = Xd.cpu().numpy()
    scale = float(np.sqrt(np.mean(ref ** 2)))
    err = float(np.sqrt(np.mean((got[:uniq] - ref) ** 2)))
    assert err / scale <= TOL, (err, scale)
    # inverse of the oracle's coefficients, and the frames beyond the checked ones through the round trip property:
    # imdct(mdct(x)) is x plus its time-domain alias, i.e. a fixed linear map -> compare against the oracle on a few rows
    Xin = torch.from_numpy(ref.astype(np.float32)).to(dev)
    xo = torch.empty(uniq, n, dtype=torch.float32, device=dev)
    m.inverse(Xin, xo)
    ref_x = np.stack([oracle.imdct(2, row.astype(np.float32)) for row in ref])
    rms_check(xo.cpu().numpy(), ref_x, f"imdct batch n={n}")
    # every frame: the same map applied to all rows (rows are independent)
    full = torch.empty(count, n, dtype=torch.float32, device=dev)
    m.inverse(Xd, full)
    y = full.cpu().numpy()
    alias = np.concatenate([x[:, :n // 2] - x[:, n // 2 - 1::-1][:, :n // 2], x[:, n // 2:] + x[:, :n // 2 - 1:-1]], axis=1)
    assert np.abs(y - alias).max() <= 2e-5 * max(1.0, np.abs(alias).max()), "imdct(mdct(x)) must be x with its time alias"
    # host buffers
    Xh = np.zeros((count, n // 2), dtype=np.float32)
    m.forward(x, Xh)
    assert np.array_equal(Xh, got)
    m.close()


@pytest.mark.parametrize("frame_len,win,channels,calls", [(128, 0, 5, (7, 1, 4)), (256, 1, 3, (1, 1, 6)), (512, 0, 70, (5, 2)),
                                                          (1024, 1, 2, (9,)), (2048, 0, 3, (3, 4)), (4096, 1, 2, (2, 3))])
def test_mdct_frames_mc_vs_oracle(dev, oracle, frame_len, win, channels, calls):
    """windowed 50 %-overlap MDCT frames in batch (llz_mdct_frames_mc_*, the register MDCT kernels with the window and the hop
    fused in): analysis and synthesis streamed over several calls of even, odd and single frame counts (the previous frame /
    the overlap-add tail carried by the handle; even and odd frames of the synthesis are separate launches), all six
    transform sizes, both windows; against the oracle's sequential handles per channel, and the TDAC property on every
    channel: the output is the input delayed by one frame"""
    F = frame_len
    total = sum(calls)
    x = oracle.synth_f32(channels, total * F, seed=F + channels)
    m = filters.MdctFramesMC(channels, F, win)
    Xs, ys, o = [], [], 0
    for frames in calls:
        xd = torch.from_numpy(np.ascontiguousarray(x[:, o * F:(o + frames) * F])).to(dev)
        Xd = torch.empty(channels, frames, F, dtype=torch.float32, device=dev)
        yd = torch.empty(channels, frames * F, dtype=torch.float32, device=dev)
        m.analysis(xd, Xd)
        m.synthesis(Xd, yd)
        Xs.append(Xd.cpu().numpy()); ys.append(yd.cpu().numpy())
        o += frames
    X, y = np.concatenate(Xs, axis=1), np.concatenate(ys, axis=1)
    for c in sorted({0, channels // 2, channels - 1}):
        Xr, yr = oracle.mdct_frames(F, win, x[c].astype(np.float64))
        assert np.sqrt(np.mean((X[c] - Xr) ** 2)) <= TOL * max(1.0, np.sqrt(np.mean(Xr ** 2))), (F, c)
        assert np.sqrt(np.mean((y[c] - yr) ** 2)) <= TOL, (F, c)
    assert np.abs(y[:, F:] - x[:, :-F]).max() <= 2e-5 and np.abs(y[:, :F]).max() <= 2e-5
    # host buffers through the staging path: a second handle, one call
    m2 = filters.MdctFramesMC(channels, F, win)
    Xh = np.zeros((channels, total, F), dtype=np.float32)
    m2.analysis(np.ascontiguousarray(x), Xh)
    assert np.array_equal(Xh, X)
    m.close(); m2.close()
    with pytest.raises(capi.LlzError):
        filters.MdctFramesMC(2, 96, 0)


@pytest.mark.parametrize("frame_len,run,channels,calls", [(128, 3, 5, (7, 1, 4, 2)), (256, 1, 3, (5, 1)), (512, 4, 9, (8, 3)),
                                                          (1024, 2, 2, (5, 6)), (128, 16, 40, (33, 16))])
def test_mdct_frames_synthesis_in_runs(dev, oracle, frame_len, run, channels, calls):
    """the synthesis with a group per RUN of consecutive segments (k_mdct_reg_f32<..., 2>: the tail of the frame before stays
    in registers, every output is written once; taken by itself for large batches, forced here): run lengths that divide the
    frame count and that do not, a single segment per group, calls of one frame; the same coefficients through the
    two-launch form give the same samples to float32 rounding, and both the oracle's"""
    F = frame_len
    total = sum(calls)
    x = oracle.synth_f32(channels, total * F, seed=F + run)
    a = filters.MdctFramesMC(channels, F, 0)
    Xd = torch.empty(channels, total, F, dtype=torch.float32, device=dev)
    a.analysis(torch.from_numpy(x).to(dev), Xd)
    a.close()
    outs = {}
    for forced in (run, 0):
        capi.tune("mdct_run", forced)
        try:
            m = filters.MdctFramesMC(channels, F, 0)
            ys, o = [], 0
            for frames in calls:
                yd = torch.full((channels, frames * F), 7.0, dtype=torch.float32, device=dev)
                m.synthesis(Xd[:, o:o + frames].contiguous(), yd)
                ys.append(yd.cpu().numpy())
                o += frames
            m.close()
        finally:
            capi.tune("mdct_run", -1)
        outs[forced] = np.concatenate(ys, axis=1)
    # (two float32 kernels: the same sums, not the same roundings)
    assert np.abs(outs[run] - outs[0]).max() <= 2e-6, "both forms add the same two windowed terms per sample"
    for c in sorted({0, channels - 1}):
        _, yr = oracle.mdct_frames(F, 0, x[c].astype(np.float64))
        assert np.sqrt(np.mean((outs[run][c] - yr) ** 2)) <= TOL, (F, c)


# ------------------------------------------------------------------------------------------------ overlap-save, 2048 points
@pytest.mark.parametrize("taps_n,channels,n", [(258, 3, 5000), (513, 5, 1536 * 4 + 1), (1025, 2, 1024 * 7), (300, 9, 700),
                                               (1025, 70, 1024 * 40 + 3), (2, 3, 4096), (513, 6, 3072 * 20 + 77),
                                               (400, 300, 3072 * 3), (512, 1, 3072 * 17)])
def test_fir_ols2048_vs_oracle_streaming(dev, oracle, taps_n, channels, n):
    """filters of 258..1025 taps on 2048-point overlap-save (k_fir_ols2k_chain_f32: one radix-2 step across the half-waves of a
    wave around the 1024-point machinery; 512 samples of overlap up to 513 taps, 1024 above): two frames (history carried by
    the handle) and the flush tail, ragged lengths, segments of several jobs, blocks that end past the frame, and the
    automatic choice"""
    taps = oracle.fir_design(po.LPF, taps_n, 0.2, 0.0, po.KAISER)
    x = oracle.synth_f32(channels, 2 * n, seed=taps_n)
    ref = oracle.fir_batch_f32(x, taps.astype(np.float32).astype(np.float64))
    f = filters.FirFilterMC(channels, n, taps, algo=filters.FIR_ALGO_OVERLAP_SAVE_2048)
    assert f.algo == filters.FIR_ALGO_OVERLAP_SAVE_2048
    outs = []
    for o in (0, n):
        xd = torch.from_numpy(np.ascontiguousarray(x[:, o:o + n])).to(dev)
        yd = torch.empty_like(xd)
        f.filter(xd, yd)
        outs.append(yd.cpu().numpy())
    tail = torch.empty(channels, taps_n - 1, dtype=torch.float32, device=dev)
    f.flush(tail)                                                  # flt_len - 1 outputs of the zero-padded stream
    f.close()
    rms_check(np.concatenate(outs, axis=1), ref, f"fir ols2048 taps={taps_n}")
    full = oracle.fir_batch_f32(np.concatenate([x, np.zeros((channels, taps_n - 1), np.float32)], axis=1),
                                taps.astype(np.float32).astype(np.float64))
    assert np.sqrt(np.mean((tail.cpu().numpy() - full[:, 2 * n:]) ** 2)) <= TOL
    if 257 < taps_n <= 513:
        g = filters.FirFilterMC(channels, n, taps)                 # AUTO: 258..513 taps
        assert g.algo == filters.FIR_ALGO_OVERLAP_SAVE_2048
        g.close()


@pytest.mark.parametrize("taps_n,channels,n", [(1026, 3, 9000), (2049, 2, 2048 * 5 + 1), (3073, 2, 1024 * 9), (1500, 40, 1024 * 30 + 3),
                                               (2, 3, 8192), (400, 5, 1000), (2049, 7, 4096 * 19 + 77), (2000, 300, 4096 * 3),
                                               (2050, 3, 4096 * 4), (1025, 4, 6144 * 3 + 5), (700, 33, 6144 * 11), (513, 6, 7168 * 5 + 77),
                                               (2600, 5, 2048 * 9 + 31), (3073, 130, 2048 * 6),
                                               # the in-between overlaps of the ladder: 768, 1536, 2560
                                               (769, 4, 6656 * 3 + 17), (600, 40, 6656 * 5), (1537, 3, 5120 * 4 + 9), (1100, 50, 5120 * 6),
                                               (2561, 2, 3072 * 7 + 100), (2100, 33, 3072 * 8)])
def test_fir_ols4096_vs_oracle_streaming(dev, oracle, taps_n, channels, n):
    """filters of up to 3073 taps on the 4096-point overlap-save (k_fir_ols4k_f32, a whole wave per pair of blocks; overlap
    512 / 1024 / 2048 / 3072 by tap count): two frames, ragged lengths, blocks that end past the frame, and the automatic
    choice beyond 513 taps"""
    taps = oracle.fir_design(po.LPF, taps_n, 0.2, 0.0, po.KAISER)
    x = oracle.synth_f32(channels, 2 * n, seed=taps_n)
    ref = oracle.fir_batch_f32(x, taps.astype(np.float32).astype(np.float64))
    f = filters.FirFilterMC(channels, n, taps, algo=filters.FIR_ALGO_OVERLAP_SAVE_4096)
    assert f.algo == filters.FIR_ALGO_OVERLAP_SAVE_4096
    outs = []
    for o in (0, n):
        xd = torch.from_numpy(np.ascontiguousarray(x[:, o:o + n])).to(dev)
        yd = torch.empty_like(xd)
        f.filter(xd, yd)
        outs.append(yd.cpu().numpy())
    tail = torch.empty(channels, taps_n - 1, dtype=torch.float32, device=dev)
    f.flush(tail)                                                  # flt_len - 1 outputs of the zero-padded stream
    f.close()
    rms_check(np.concatenate(outs, axis=1), ref, f"fir ols4096 taps={taps_n}")
    full = oracle.fir_batch_f32(np.concatenate([x, np.zeros((channels, taps_n - 1), np.float32)], axis=1),
                                taps.astype(np.float32).astype(np.float64))
    assert np.sqrt(np.mean((tail.cpu().numpy() - full[:, 2 * n:]) ** 2)) <= TOL
    if taps_n > 513:
        g = filters.FirFilterMC(channels, n, taps)                 # AUTO: 514..1025 taps here, the 8192-point form beyond
        assert g.algo == (filters.FIR_ALGO_OVERLAP_SAVE_4096 if taps_n <= 1025 else filters.FIR_ALGO_OVERLAP_SAVE_8192)
        g.close()


@pytest.mark.parametrize("taps_n,channels,n", [(3073, 2, 10240 * 3 + 5), (3073, 70, 10240 * 2), (2561, 3, 11264 * 2 + 100), (2049, 5, 12288 * 3),
                                               (1537, 4, 13312 * 2 + 1), (4097, 2, 8192 * 3 + 9), (3585, 9, 9216 * 4), (3000, 300, 10240 * 3),
                                               (2, 3, 20000), (1000, 7, 5000), (4097, 33, 8192 * 5 + 4000), (3073, 1100, 10240 * 2 + 17),
                                               # frames shorter than the overlap (the history supplies most of a block), one channel
                                               (3073, 3, 1700), (2100, 1, 5000), (4097, 2, 2500),
                                               # more overlap than new samples: block B starts in the history too, rows 64.. are overlap
                                               (6145, 2, 4096 * 3 + 50), (5000, 40, 6144 * 2), (5121, 3, 3000), (6145, 1, 2000)])
def test_fir_ols8192_vs_oracle_streaming(dev, oracle, taps_n, channels, n):
    """filters of up to 6145 taps on the 8192-point overlap-save (k_fir_ols8k_f32: a PAIR of waves per pair of blocks, the halves
    of the radix-2 step swapped through LDS under the pair's own round counters; overlap 1536 ... 6144 by tap count): two frames
    (the second starts from the first's history), ragged lengths, blocks that end past the frame, more pairs of waves than
    segments and fewer.  Large batches are checked on a spread of 24 channels (the oracle is one core)."""
    taps = oracle.fir_design(po.LPF, taps_n, 0.2, 0.0, po.KAISER)
    x = oracle.synth_f32(channels, 2 * n, seed=taps_n)
    sel = np.arange(channels) if channels <= 40 else np.unique(np.linspace(0, channels - 1, 24).astype(int))
    t64 = taps.astype(np.float32).astype(np.float64)
    ref = oracle.fir_batch_f32(np.ascontiguousarray(x[sel]), t64)
    f = filters.FirFilterMC(channels, n, taps, algo=filters.FIR_ALGO_OVERLAP_SAVE_8192)
    assert f.algo == filters.FIR_ALGO_OVERLAP_SAVE_8192
    outs = []
    for o in (0, n):
        xd = torch.from_numpy(np.ascontiguousarray(x[:, o:o + n])).to(dev)
        yd = torch.empty_like(xd)
        f.filter(xd, yd)
        outs.append(yd.cpu().numpy()[sel])
    tail = torch.empty(channels, taps_n - 1, dtype=torch.float32, device=dev)
    f.flush(tail)
    f.close()
    rms_check(np.concatenate(outs, axis=1), ref, f"fir ols8192 taps={taps_n}")
    full = oracle.fir_batch_f32(np.concatenate([x[sel], np.zeros((len(sel), taps_n - 1), np.float32)], axis=1), t64)
    assert np.sqrt(np.mean((tail.cpu().numpy()[sel] - full[:, 2 * n:]) ** 2)) <= TOL


# ------------------------------------------------------------------------------------------------ overlap-save, chain form
def test_fir_ols_small_and_ragged_batches(dev, oracle):
    """the 1024-point overlap-save kernel on batches far smaller than the headline (test_fir_ols_headline_shape_full_length
    does the large one): ragged lengths, segments shorter than 16 jobs, fewer segments than half-wave slots, odd row lengths
    and streaming across calls"""
    for channels, n, taps_n in ((3, 1536 * 40 + 100, 257), (5, 1536 * 33, 63), (2, 1000, 129), (9, 1536 * 17 + 1, 200),
                                (4, 1536 * 5 + 7, 257), (700, 1536 * 3, 99)):
        taps = oracle.fir_design(po.LPF, taps_n, 0.2, 0.0, po.KAISER)
        x = oracle.synth_f32(channels, 2 * n, seed=taps_n)
        ref = oracle.fir_batch_f32_mt(x, taps.astype(np.float32).astype(np.float64))
        f = filters.FirFilterMC(channels, n, taps, algo=filters.FIR_ALGO_OVERLAP_SAVE)
        outs = []
        for o in (0, n):                                    # two frames: the history carried between calls
            xd = torch.from_numpy(np.ascontiguousarray(x[:, o:o + n])).to(dev)
            yd = torch.empty_like(xd)
            f.filter(xd, yd)
            outs.append(yd.cpu().numpy())
        f.close()
        rms_check(np.concatenate(outs, axis=1), ref, f"fir ols {channels}x{n}x{taps_n}")


def test_fir_ols_headline_shape_full_length(dev, oracle):
    """BASELINE config 3 at its full size through the DEFAULT launcher (chain form, wide accesses): 4096 channels x 2^20
    samples x 257 taps.  Eight channels are checked over their FULL length against the oracle (first, last and six spread
    so that they fall into different rounds of the persistent grid; every one crosses all 42 segment boundaries and ends
    in the ragged 11-job last segment), plus every channel over its first 16 Ki samples (SURVEY.md 8(d) parity measure)."""
    channels, n, taps_n = 4096, 1 << 20, 257
    taps = oracle.fir_design(po.LPF, taps_n, 0.1, 0.0, po.KAISER)
    x = torch.empty(channels, n, dtype=torch.float32, device=dev)
    filters.synth_f32(x, seed=0x11C0FFEE)
    y = torch.empty_like(x)
    f = filters.FirFilterMC(channels, n, taps)
    assert f.algo == filters.FIR_ALGO_OVERLAP_SAVE
    f.filter(x, y)
    torch.cuda.synchronize()
    h = taps.astype(np.float32).astype(np.float64)
    sel = [0, 1, 95, 1337, 2048, 3071, 4094, 4095]
    ref = oracle.fir_batch_f32_mt(x[sel].cpu().numpy(), h)
    rms_check(y[sel].cpu().numpy(), ref, "fir ols headline, 8 channels x 2^20")
    m = 1 << 14
    ref = oracle.fir_batch_f32_mt(x[:, :m].cpu().numpy(), h)
    rms_check(y[:, :m].cpu().numpy(), ref, "fir ols headline, all channels x 16 Ki")
    # second call: streaming state (the previous call's last 256 samples) through the same launch shape
    f.filter(x, y)
    torch.cuda.synchronize()
    xs = torch.cat([x[sel, n - (taps_n - 1):], x[sel, :m]], dim=1).cpu().numpy()
    ref = oracle.fir_batch_f32(xs, h)[:, taps_n - 1:]
    rms_check(y[sel, :m].cpu().numpy(), ref, "fir ols headline, second call")
    f.close()


@pytest.mark.parametrize("name,row,prec", [("config 4 low-pass section", [0.2066, 0.4131, 0.2066, 1.0, -0.3695, 0.1958], 32),
                                           ("0.99-radius set", [0.01, 0.0, -0.01, 1.0, -2 * 0.99 * np.cos(0.3), 0.99 ** 2], 64)])
def test_iir_config4_full_size(dev, oracle, name, row, prec):
    """BASELINE config 4 at its full size through the handle's own dispatch: 1024 channels x 2^20 samples x 8 sections, both
    coefficient sets the bench times (packed float32 with 32 samples per lane; double with 32 samples per lane, b0 folded).
    Six spread channels over their FULL length against the double oracle, every channel over its first 8 Ki samples, and a
    second call of a ragged length (state handed from the 2048-sample kernels to the 1024-sample and per-channel ones)"""
    channels, n = 1024, 1 << 20
    coef = np.tile(np.array(row), (8, 1))
    x = torch.empty(channels, n, dtype=torch.float32, device=dev)
    filters.synth_f32(x, seed=0x44)
    y = torch.empty_like(x)
    f = filters.IirCascadeMC(channels, coef)
    assert f.precision == prec
    f.filter(x, y)
    torch.cuda.synchronize()
    sel = [0, 1, 333, 512, 1022, 1023]
    n2 = 1024 * 3 + 77
    x2 = torch.empty(channels, n2, dtype=torch.float32, device=dev)
    filters.synth_f32(x2, seed=0x45)
    y2 = torch.empty_like(x2)
    f.filter(x2, y2)
    f.close()
    xs = np.concatenate([x[sel].cpu().numpy(), x2[sel].cpu().numpy()], axis=1)
    ref = oracle.iir_cascade_batch_f32(xs, coef)
    got = np.concatenate([y[sel].cpu().numpy(), y2[sel].cpu().numpy()], axis=1).astype(np.float64)
    err, scale = float(np.sqrt(np.mean((got - ref) ** 2))), float(np.sqrt(np.mean(ref ** 2)))
    assert err <= TOL * max(1.0, scale) and err / scale <= TOL, (name, err, scale)
    m = 8192
    ref = oracle.iir_cascade_batch_f32(x[:, :m].cpu().numpy(), coef)
    got = y[:, :m].cpu().numpy().astype(np.float64)
    err, scale = float(np.sqrt(np.mean((got - ref) ** 2))), float(np.sqrt(np.mean(ref ** 2)))
    assert err <= TOL * max(1.0, scale) and err / scale <= TOL, (name, "all channels", err, scale)


def test_resample_config5_full_size_i16_bit_exact(dev, oracle):
    """BASELINE config 5 at its full size in the reference's own sample format: 8192 channels x 4 Mi int16 samples, 1:3,
    through the handle's own dispatch (integer screen on the matrix cores + exact recompute).  Six spread channels over their
    FULL length and every channel over its first 12 Ki inputs, bit for bit against the oracle; a checksum of the whole output
    must not change when the same frame is processed by a second handle on the all-double kernel for a spread subset"""
    channels, n_in, M = 8192, 1536 * 2730, 3                            # whole frames of the reference (1536 samples), ~4 Mi
    x = torch.empty(channels, n_in, dtype=torch.int16, device=dev)
    filters.synth_i16(x, seed=0x55)
    r = filters.ResampleMC(channels, 1, M, 1.0, po.BLACKMAN, filters.PCM_I16)
    n_out = r.out_len(n_in)
    y = torch.empty(channels, n_out, dtype=torch.int16, device=dev)
    r.process(x, y)
    torch.cuda.synchronize()
    r.close()
    sel = [0, 1, 2047, 4096, 8190, 8191]
    ref = oracle.rs_batch_i16(x[sel].cpu().numpy(), 1, M, 1.0, po.BLACKMAN)
    assert np.array_equal(y[sel].cpu().numpy(), ref), "config 5 int16: six channels, full length"
    m_in = 3 * 4096
    ref = oracle.rs_batch_i16(x[:, :m_in].cpu().numpy(), 1, M, 1.0, po.BLACKMAN)
    assert np.array_equal(y[:, :m_in // M].cpu().numpy(), ref), "config 5 int16: all channels x 4 Ki outputs"
    # the all-double kernel on 64 spread channels, whole length: same bits
    sub = list(range(0, channels, 128))
    xs = x[sub].contiguous()
    with capi.tuned(rs_i16_path=1):
        r2 = filters.ResampleMC(len(sub), 1, M, 1.0, po.BLACKMAN, filters.PCM_I16)
        y2 = torch.empty(len(sub), n_out, dtype=torch.int16, device=dev)
        r2.process(xs, y2)
        r2.close()
    assert torch.equal(y2, y[sub])


def test_resample_config5_full_size_f32(dev, oracle):
    """BASELINE config 5 at its full size, float32: 8192 channels x ~4 Mi samples, 1:3 on the matrix cores (split bf16).  Four
    spread channels over their FULL length and every channel over its first 12 Ki inputs against the oracle (double
    accumulate over the float-rounded taps)"""
    channels, n_in, M = 8192, 1536 * 2730, 3
    x = torch.empty(channels, n_in, dtype=torch.float32, device=dev)
    filters.synth_f32(x, seed=0x56)
    r = filters.ResampleMC(channels, 1, M, 1.0, po.BLACKMAN, filters.PCM_F32)
    n_out = r.out_len(n_in)
    y = torch.empty(channels, n_out, dtype=torch.float32, device=dev)
    r.process(x, y)
    torch.cuda.synchronize()
    r.close()
    sel = [0, 2049, 6000, 8191]
    ref = oracle.rs_batch_f32(x[sel].cpu().numpy(), 1, M, 1.0, po.BLACKMAN)
    got = y[sel].cpu().numpy().astype(np.float64)
    err = float(np.sqrt(np.mean((got - ref) ** 2)))
    assert err <= TOL and err / float(np.sqrt(np.mean(ref ** 2))) <= TOL, err
    m_in = 3 * 4096
    ref = oracle.rs_batch_f32(x[:, :m_in].cpu().numpy(), 1, M, 1.0, po.BLACKMAN)
    got = y[:, :m_in // M].cpu().numpy().astype(np.float64)
    err = float(np.sqrt(np.mean((got - ref) ** 2)))
    assert err <= TOL and err / float(np.sqrt(np.mean(ref ** 2))) <= TOL, err


def test_iir_forced_segment_count_is_clamped(dev, oracle):
    """llz_hip_tune("iir_segs", n) forces the time-segment count for A/B runs; a count that would leave a segment shorter than
    its own warm-up (its first chunk would lie in front of the row) is reduced by every launcher"""
    rows = []
    for k in range(8):
        r, th = 0.99 - 0.002 * k, 0.3 + 0.2 * k
        a1, a2 = -2 * r * np.cos(th), r * r
        rows.append([(1 + a1 + a2) / 4, (1 + a1 + a2) / 2, (1 + a1 + a2) / 4, 1.0, a1, a2])
    coef = np.array(rows)
    channels, n = 2048, 1024 * 96
    x = torch.empty(channels, n, dtype=torch.float32, device=dev)
    filters.synth_f32(x, seed=11)
    sel = [0, 1000, 2047]
    ref = oracle.iir_cascade_batch_f32(x[sel].cpu().numpy(), coef)
    for tuned in ({"iir_segs": 64}, {"iir_segs": 64, "iir_unpacked": 2}, {"iir_segs": 64, "iir_pipe": 1}):
        with capi.tuned(**tuned):
            f = filters.IirCascadeMC(channels, coef)
            y = torch.empty_like(x)
            f.filter(x, y)
            f.close()
        got = y[sel].cpu().numpy().astype(np.float64)
        err, scale = float(np.sqrt(np.mean((got - ref) ** 2))), float(np.sqrt(np.mean(ref ** 2)))
        assert err <= TOL * max(1.0, scale) and err / scale <= TOL, (tuned, err, scale)


@pytest.mark.parametrize("segs", [2, 5, 16])
@pytest.mark.parametrize("form", ["pipe", "wave32", "wave64"])
def test_iir_time_segments_stream_across_calls(dev, oracle, form, segs):
    """several time segments per channel in ONE launch, streamed over two calls: segment 0 reads the frame's start state
    while the last segment of the same launch writes the frame's end state -- different workgroups that nothing orders, so
    the two states live in different buffers (the handle swaps them after the launch)"""
    radius = 0.99 if form == "wave64" else 0.5
    rows = []
    for k in range(4):
        r, th = radius - 0.02 * k, 0.4 + 0.3 * k
        a1, a2 = -2 * r * np.cos(th), r * r
        rows.append([(1 + a1 + a2) / 4, (1 + a1 + a2) / 2, (1 + a1 + a2) / 4, 1.0, a1, a2])
    coef = np.array(rows)
    channels, n = 24, 1024 * (2048 if form == "wave64" else 256)
    with capi.tuned(iir_segs=segs, iir_pipe=1 if form == "pipe" else -1, iir_wave_min_items=0):
        f = filters.IirCascadeMC(channels, coef)
        xs, ys = [], []
        for call in range(2):
            x = torch.empty(channels, n, dtype=torch.float32, device=dev)
            filters.synth_f32(x, seed=60 + call)
            y = torch.empty_like(x)
            f.filter(x, y)
            xs.append(x.cpu().numpy()); ys.append(y.cpu().numpy())
        f.close()
    ref = oracle.iir_cascade_batch_f32(np.concatenate(xs, axis=1), coef)
    got = np.concatenate(ys, axis=1).astype(np.float64)
    err, scale = float(np.sqrt(np.mean((got - ref) ** 2))), float(np.sqrt(np.mean(ref ** 2)))
    assert err <= 1e-5 * max(1.0, scale) and err / scale <= 1e-5, (form, segs, err, scale)


# ------------------------------------------------------------------------------------------------ limits and degenerate calls
def test_limits_and_degenerate_calls(dev, oracle):
    """the largest channel count the batch handles accept, one-sample frames, and refused calls: zero-length frames and
    out-of-range shapes come back as errors (negative code / (unsigned long)-1 with a message), never as a crash"""
    taps = oracle.fir_design(po.LPF, 5, 0.3, 0.0, po.HAMMING)
    # 65535 channels (the grid.y limit of the tile kernels) x 64 samples; spot-check the ends and the middle
    ch, n = 65535, 64
    x = torch.empty(ch, n, dtype=torch.float32, device=dev)
    filters.synth_f32(x[:1000], seed=3)
    x[1000:] = x[:1000].repeat(65, 1)[:ch - 1000]
    y = torch.empty_like(x)
    f = filters.FirFilterMC(ch, n, taps)
    f.filter(x, y)
    sel = [0, 1, 32767, 65533, 65534]
    ref = oracle.fir_batch_f32(x[sel].cpu().numpy(), taps.astype(np.float32).astype(np.float64))
    rms_check(y[sel].cpu().numpy(), ref, "fir 65535 channels")
    f.close()
    with pytest.raises(capi.LlzError):
        filters.FirFilterMC(65536, n, taps)
    # one-sample frames stream correctly
    f = filters.FirFilterMC(2, 1, taps)
    xs = oracle.synth_f32(2, 40, seed=9)
    out = np.zeros_like(xs)
    for i in range(40):
        xi = torch.from_numpy(np.ascontiguousarray(xs[:, i:i + 1])).to(dev)
        yi = torch.empty_like(xi)
        f.filter(xi, yi)
        out[:, i] = yi.cpu().numpy()[:, 0]
    f.close()
    rms_check(out, oracle.fir_batch_f32(xs, taps.astype(np.float32).astype(np.float64)), "fir one-sample frames")
    # refused: zero-length frame, zero channels, zero taps, absurd resample ratio, IIR without stages
    L = capi.lib()
    f = filters.FirFilterMC(2, 8, taps)
    z = torch.empty(2, 8, dtype=torch.float32, device=dev)
    assert L.llz_fir_filter_mc(f.handle, z.data_ptr(), z.data_ptr(), 0) < 0 and capi.last_error()
    f.close()
    for bad in (lambda: filters.FirFilterMC(0, 8, taps), lambda: filters.FirFilterMC(2, 8, np.zeros(0)),
                lambda: filters.ResampleMC(2, 1, 17, 1.0, po.BLACKMAN, filters.PCM_F32),
                lambda: filters.IirCascadeMC(2, np.zeros((0, 6))), lambda: filters.FftBatch(48),
                lambda: filters.StftMC(2, 0, 3), lambda: filters.MdctBatch(100)):
        with pytest.raises(capi.LlzError):
            bad()
    q = filters.IirCascadeMC(70000, np.array([[0.2066, 0.4131, 0.2066, 1.0, -0.3695, 0.1958]]))   # no channel limit here
    xi = torch.zeros(70000, 16, dtype=torch.float32, device=dev)
    xi[:, 0] = 1.0
    yi = torch.empty_like(xi)
    q.filter(xi, yi)
    q.close()
    h = oracle.iir_cascade_batch_f32(xi[:1].cpu().numpy(), np.array([[0.2066, 0.4131, 0.2066, 1.0, -0.3695, 0.1958]]))
    assert np.abs(yi[[0, 69999]].cpu().numpy() - h).max() < 1e-6


# ------------------------------------------------------------------------------------------------ sharded handles (llz_shard.h)
def _virtual(n):
    """n shards, all on device 0: how a one-GPU box runs the multi-GPU code path"""
    return [0] * n


@pytest.mark.parametrize("n_shards", [2, 8])
@pytest.mark.parametrize("algo,taps_n", [(filters.FIR_ALGO_OVERLAP_SAVE, 257), (filters.FIR_ALGO_TIME, 63),
                                         (filters.FIR_ALGO_OVERLAP_SAVE_2048, 400), (filters.FIR_ALGO_OVERLAP_SAVE_8192, 2100)])
def test_sharded_fir_equals_unsharded(dev, oracle, n_shards, algo, taps_n):
    """the C ABI's sharded FIR handle (contiguous channel ranges, one stream per shard, tables broadcast from shard 0) gives
    bit-identical output to one unsharded handle, over two streamed frames and the flush"""
    channels, n = 37, 1536 * 9 + 5
    taps = oracle.fir_design(po.LPF, taps_n, 0.15, 0.0, po.KAISER)
    x = torch.empty(channels, 2 * n, dtype=torch.float32, device=dev)
    filters.synth_f32(x, seed=21)
    ref_h = filters.FirFilterMC(channels, n, taps, algo=algo)
    sh = filters.FirFilterMCSharded(channels, n, taps, _virtual(n_shards), algo=algo)
    assert [c for (_d, _c0, c) in sh.shards] == [channels // n_shards + (1 if s < channels % n_shards else 0) for s in range(n_shards)]
    for o in (0, n):
        xin = x[:, o:o + n].contiguous()
        y_ref = torch.empty_like(xin)
        ref_h.filter(xin, y_ref)
        y = torch.full_like(xin, float("nan"))
        torch.cuda.synchronize()          # the shards' streams are non-blocking: they do not wait for torch's stream
        sh.filter(sh.split(xin), sh.split(y))
        sh.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(y, y_ref), f"frame at {o}"
    t_ref = torch.empty(channels, taps_n - 1, dtype=torch.float32, device=dev)
    ref_h.flush(t_ref)
    t = torch.empty_like(t_ref)
    torch.cuda.synchronize()
    assert sh.flush(sh.split(t)) == taps_n - 1
    sh.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(t, t_ref)
    rms_check(y_ref.cpu().numpy(),
              oracle.fir_batch_f32(x.cpu().numpy(), taps.astype(np.float32).astype(np.float64))[:, n:], "unsharded vs oracle")
    ref_h.close(); sh.close()


@pytest.mark.parametrize("n_shards", [2, 8])
@pytest.mark.parametrize("L,M,fmt", [(1, 3, filters.PCM_F32), (1, 3, filters.PCM_I16), (147, 160, filters.PCM_F32), (2, 3, filters.PCM_I16)])
def test_sharded_resample_equals_unsharded(dev, oracle, n_shards, L, M, fmt):
    channels = 19
    n_in = M * 64 * (5 if L > 4 else 40)
    n_out = n_in * L // M
    dt = torch.float32 if fmt == filters.PCM_F32 else torch.int16
    x = torch.empty(channels, 2 * n_in, dtype=dt, device=dev)
    (filters.synth_f32 if fmt == filters.PCM_F32 else filters.synth_i16)(x, seed=5)
    ref_h = filters.ResampleMC(channels, L, M, 1.0, po.BLACKMAN, fmt)
    sh = filters.ResampleMCSharded(channels, L, M, 1.0, po.BLACKMAN, fmt, _virtual(n_shards))
    for o in (0, n_in):
        xin = x[:, o:o + n_in].contiguous()
        y_ref = torch.empty(channels, n_out, dtype=dt, device=dev)
        ref_h.process(xin, y_ref)
        y = torch.zeros_like(y_ref)
        torch.cuda.synchronize()
        assert sh.process(sh.split(xin), sh.split(y)) == n_out
        sh.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(y, y_ref), f"frame at {o}"
    ref_h.close(); sh.close()


@pytest.mark.parametrize("n_shards", [2, 8])
@pytest.mark.parametrize("radius", [0.44, 0.99])
def test_sharded_iir_equals_unsharded(dev, oracle, n_shards, radius):
    """same kernels on channel shards: the kernel form and the time-segment count are shape dependent, so both runs are
    pinned to one form (one segment per channel, stage pipeline) before comparing bit for bit"""
    rows = []
    for k in range(8):
        r, th = radius - 0.01 * k, 0.3 + 0.2 * k
        a1, a2 = -2 * r * np.cos(th), r * r
        rows.append([(1 + a1 + a2) / 4, (1 + a1 + a2) / 2, (1 + a1 + a2) / 4, 1.0, a1, a2])
    coef = np.array(rows)
    channels, n = 21, 1024 * 12 + 36
    x = torch.empty(channels, 2 * n, dtype=torch.float32, device=dev)
    filters.synth_f32(x, seed=77)
    with capi.tuned(iir_segs=1, iir_pipe=1):
        ref_h = filters.IirCascadeMC(channels, coef)
        sh = filters.IirCascadeMCSharded(channels, coef, _virtual(n_shards))
        for o in (0, n):
            xin = x[:, o:o + n].contiguous()
            y_ref = torch.empty_like(xin)
            ref_h.filter(xin, y_ref)
            y = torch.zeros_like(xin)
            torch.cuda.synchronize()
            sh.filter(sh.split(xin), sh.split(y))
            sh.synchronize()
            torch.cuda.synchronize()
            assert torch.equal(y, y_ref), f"frame at {o}"
        ref_h.close(); sh.close()
    ref = oracle.iir_cascade_batch_f32(x.cpu().numpy(), coef)[:, n:]
    err = float(np.sqrt(np.mean((y_ref.cpu().numpy() - ref) ** 2)))
    assert err <= 1e-5 * max(1.0, float(np.sqrt(np.mean(ref ** 2))))


def test_sharded_tables_through_rccl_on_one_device(dev, oracle):
    """the table broadcast runs through RCCL itself (dlopen of librccl, ncclCommInitAll, grouped ncclBroadcast on the shard's
    stream, ncclCommDestroy) when llz_hip_tune("shard_rccl", 1) asks for it on a single device: a one-rank communicator,
    the same calls a multi-GPU node makes; further shards on the device take device-to-device copies"""
    channels, n, taps_n = 12, 4000, 129
    taps = oracle.fir_design(po.LPF, taps_n, 0.2, 0.0, po.HAMMING)
    x = torch.empty(channels, n, dtype=torch.float32, device=dev)
    filters.synth_f32(x, seed=3)
    ref_h = filters.FirFilterMC(channels, n, taps)
    y_ref = torch.empty_like(x)
    ref_h.filter(x, y_ref)
    with capi.tuned(shard_rccl=1):
        sh = filters.FirFilterMCSharded(channels, n, taps, _virtual(3))
    y = torch.zeros_like(x)
    torch.cuda.synchronize()
    sh.filter(sh.split(x), sh.split(y))
    sh.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    y2 = torch.zeros_like(x)
    torch.cuda.synchronize()
    sh.timer_start()
    sh.filter(sh.split(x), sh.split(y2))
    sh.timer_stop()
    ms, per = sh.timer_ms()
    assert ms > 0 and per.shape == (3,) and ms == per.max()
    ref_h.close(); sh.close()


def _per_device(sh, whole):
    """copies of a [channels, n] tensor's shard rows, each on its shard's GPU"""
    return [whole[c0:c0 + cnt].to(torch.device("cuda", d)) for (d, c0, cnt) in sh.shards]


def test_sharded_over_distinct_devices(dev, oracle):
    """the library's own multi-GPU path where a node offers it: llz_*_mc_sharded_init over DISTINCT devices (one shard per
    visible GPU: ncclCommInitAll + one grouped ncclBroadcast per table), every shard's buffers on its own GPU, against one
    unsharded handle on GPU 0 -- bit for bit, FIR / resample / IIR.  Also: a buffer that lives on the wrong GPU is refused
    with LLZ_ERR_ARG (no peer access is ever enabled).  Skipped on a one-GPU box."""
    ndev = capi.lib().llz_hip_device_count()
    if ndev < 2:
        pytest.skip("one GPU visible: the distinct-device path needs two")
    devices = list(range(ndev))
    L = capi.lib()
    channels, n = 8 * ndev + 3, 1536 * 6
    taps = oracle.fir_design(po.LPF, 257, 0.1, 0.0, po.KAISER)
    x = torch.empty(channels, n, dtype=torch.float32, device=dev)
    filters.synth_f32(x, seed=31)
    torch.cuda.synchronize()
    # FIR
    ref_h = filters.FirFilterMC(channels, n, taps)
    y_ref = torch.empty_like(x)
    ref_h.filter(x, y_ref)
    sh = filters.FirFilterMCSharded(channels, n, taps, devices)
    assert sh.rccl_ranks == ndev and [d for (d, _c0, _c) in sh.shards] == devices
    xs, ys = _per_device(sh, x), sh.alloc(n, torch.float32)
    sh.filter(xs, ys)
    sh.synchronize()
    assert L.llz_hip_get_device() == 0                                   # the caller's device is restored
    assert torch.equal(torch.cat([y.to(dev) for y in ys]), y_ref)
    with pytest.raises(capi.LlzError, match="lives on GPU"):
        sh.filter([xs[0]] * ndev, ys)                                    # shard 1 handed a buffer of GPU 0
    with pytest.raises(capi.LlzError):
        sh.split(x)                                                      # one tensor cannot serve shards on several GPUs
    ref_h.close(); sh.close()
    # resample 1:3, float32 and the reference's int16
    n_in = 3 * 4096
    for fmt, dt, synth in ((filters.PCM_F32, torch.float32, filters.synth_f32), (filters.PCM_I16, torch.int16, filters.synth_i16)):
        xr = torch.empty(channels, n_in, dtype=dt, device=dev)
        synth(xr, seed=32)
        torch.cuda.synchronize()
        ref_r = filters.ResampleMC(channels, 1, 3, 1.0, po.BLACKMAN, fmt)
        yr_ref = torch.empty(channels, n_in // 3, dtype=dt, device=dev)
        ref_r.process(xr, yr_ref)
        shr = filters.ResampleMCSharded(channels, 1, 3, 1.0, po.BLACKMAN, fmt, devices)
        xs, ys = _per_device(shr, xr), shr.alloc(n_in // 3, dt)
        assert shr.process(xs, ys) == n_in // 3
        shr.synchronize()
        assert torch.equal(torch.cat([y.to(dev) for y in ys]), yr_ref), fmt
        ref_r.close(); shr.close()
    # IIR (both runs pinned to one kernel form: the form is shape dependent)
    coef = np.tile(np.array([0.2066, 0.4131, 0.2066, 1.0, -0.3695, 0.1958]), (8, 1))
    with capi.tuned(iir_segs=1, iir_pipe=1):
        ref_q = filters.IirCascadeMC(channels, coef)
        yq_ref = torch.empty_like(x)
        ref_q.filter(x, yq_ref)
        shq = filters.IirCascadeMCSharded(channels, coef, devices)
        xs, ys = _per_device(shq, x), shq.alloc(n, torch.float32)
        shq.filter(xs, ys)
        shq.synchronize()
        assert torch.equal(torch.cat([y.to(dev) for y in ys]), yq_ref)
        ref_q.close(); shq.close()
    torch.cuda.synchronize()


def test_sharded_refusals(dev, oracle):
    taps = oracle.fir_design(po.LPF, 33, 0.2, 0.0, po.HAMMING)
    for devices in ([], [0, 99], [0] * 65, [-1]):
        with pytest.raises(capi.LlzError):
            filters.FirFilterMCSharded(8, 256, taps, devices)
    with pytest.raises(capi.LlzError):
        filters.FirFilterMCSharded(2, 256, taps, [0, 0, 0])    # more shards than channels


def test_handle_binds_its_device(dev, oracle):
    """a handle records the device it was created on and binds it in every call (the caller's current device is restored).
    On a one-GPU box this only shows that binding does not disturb anything; test_sharded_over_distinct_devices is the real
    check where two GPUs are visible."""
    L = capi.lib()
    taps = oracle.fir_design(po.LPF, 33, 0.2, 0.0, po.HAMMING)
    f = filters.FirFilterMC(3, 512, taps)
    x = torch.empty(3, 512, dtype=torch.float32, device=dev)
    filters.synth_f32(x, seed=1)
    y = torch.empty_like(x)
    f.filter(x, y)
    assert L.llz_hip_get_device() == 0
    f.close()


# ------------------------------------------------------------------------------------------------ int16 decimator, screened on the matrix cores
@pytest.mark.parametrize("M,win,gain", [(3, po.BLACKMAN, 1.0), (2, po.HAMMING, 1.0), (5, po.KAISER, 1.0), (3, po.BLACKMAN, 2.5),
                                        (4, po.HAMMING, 0.37), (3, po.HAMMING, 100.0), (2, po.BLACKMAN, 1e-3)])
def test_resample_i16_screened_is_bit_exact(dev, oracle, M, win, gain):
    """LLZ_PCM_I16 with L = 1 runs the integer screen on the matrix cores (fir_mfma_i8.hip) and recomputes in the reference's
    double order only the outputs the screen cannot decide.  Bit-exact against the oracle on: random PCM, full-scale PCM that
    drives the clamp rails, digital silence, DC (every output ON a truncation edge: all of them take the recompute path),
    silence -> signal transitions inside a tile, streamed over two calls; and identical to the all-double kernel."""
    info = oracle.rs_info(2, 1, M, gain, win)
    nin = info["bytes_in"] // 2 * 24
    ch = 9
    x = oracle.synth_i16(ch, nin, seed=M * 19)
    x[1] = (x[1].astype(np.int32) * 2).clip(-32768, 32767).astype(np.int16)     # clipping rails
    x[2] = 0                                                                    # digital silence
    x[3] = 12345                                                                # DC: output = 12345 * sum(g) * gain, near-integer
    x[4, : nin // 2] = 0                                                        # silence, then signal
    x[5] = -32768
    x[6] = 32767
    x[7] = np.where(np.arange(nin) % 2 == 0, 1, -1) * 20000                     # alternating full-ish scale
    ref = oracle.rs_batch_i16(x, 1, M, gain, win)
    outs, outs_plain = [], []
    cut = (nin // 2) // M * M
    for tuned, dst in (({}, outs), ({"rs_i16_path": 1}, outs_plain)):
        with capi.tuned(**tuned):
            r = filters.ResampleMC(ch, 1, M, gain, win, filters.PCM_I16)
            for (o, e) in ((0, cut), (cut, nin)):
                xi = torch.from_numpy(np.ascontiguousarray(x[:, o:e])).to(dev)
                yi = torch.empty(ch, (e - o) // M, dtype=torch.int16, device=dev)
                r.process(xi, yi)
                dst.append(yi.cpu().numpy())
            r.close()
    got = np.concatenate(outs, axis=1)
    bad = np.argwhere(got != ref)
    assert bad.size == 0, f"{len(bad)} samples differ, first at {bad[0]}: got {got[tuple(bad[0])]} ref {ref[tuple(bad[0])]}"
    assert np.array_equal(np.concatenate(outs_plain, axis=1), ref)


def test_resample_i16_screened_large_batch(dev, oracle):
    """a batch large enough for the persistent grid to walk several tiles per workgroup (prefetch path, ragged last tile,
    many channels): bit-exact on a spread of channels"""
    ch, nin = 300, 1536 * 29          # whole reference frames (1536 samples at 1:3), a ragged last tile
    x = torch.empty(ch, nin, dtype=torch.int16, device=dev)
    filters.synth_i16(x, seed=11)
    y = torch.empty(ch, nin // 3, dtype=torch.int16, device=dev)
    r = filters.ResampleMC(ch, 1, 3, 1.0, po.BLACKMAN, filters.PCM_I16)
    r.process(x, y)
    sel = [0, 1, 77, 150, 298, 299]
    ref = oracle.rs_batch_i16(x[sel].cpu().numpy(), 1, 3, 1.0, po.BLACKMAN)
    assert np.array_equal(y[sel].cpu().numpy(), ref)
    r.close()


# ------------------------------------------------------------------------------------------------ int16 L/M resampler, screened per phase
@pytest.mark.parametrize("L,M,win,gain", [(147, 160, po.BLACKMAN, 1.0), (160, 147, po.BLACKMAN, 1.0), (2, 3, po.HAMMING, 1.0),
                                          (3, 2, po.KAISER, 1.0), (20, 147, po.BLACKMAN, 1.0), (147, 160, po.HAMMING, 2.5),
                                          (160, 147, po.KAISER, 0.37), (7, 5, po.BLACKMAN, 1e-3), (441, 320, po.BLACKMAN, 1.0)])
def test_resample_i16_lm_screened_is_bit_exact(dev, oracle, L, M, win, gain):
    """LLZ_PCM_I16 with L >= 2 runs the per-phase integer screen on the matrix cores (resample_i8.hip: the reference CLI's own
    ratios 147:160 and 160:147 among them) and recomputes in the reference's double order only the outputs the screen cannot
    decide.  Bit-exact against the oracle's llz_resample loop on random PCM, full-scale PCM on the clamp rails, digital silence,
    DC (outputs ON truncation edges), silence -> signal inside a span, both rails and an alternating signal; streamed over two
    calls (history in front of the second); and identical to the all-double kernel."""
    info = oracle.rs_info(2, L, M, gain, win)
    nin1 = info["bytes_in"] // 2
    frames = max(2, min(12, 60000 // nin1))
    nin = nin1 * frames
    ch = 9
    x = oracle.synth_i16(ch, nin, seed=L * 7 + M)
    x[1] = (x[1].astype(np.int32) * 2).clip(-32768, 32767).astype(np.int16)     # clipping rails
    x[2] = 0                                                                    # digital silence
    x[3] = 12345                                                                # DC
    x[4, : nin // 2] = 0                                                        # silence, then signal
    x[5] = -32768
    x[6] = 32767
    x[7] = np.where(np.arange(nin) % 2 == 0, 1, -1) * 20000                     # alternating
    ref = oracle.rs_batch_i16(x, L, M, gain, win)
    outs, outs_plain = [], []
    cut = (frames // 2) * nin1                                                  # whole reference frames: a period boundary
    outs_first = []
    # the default launch (k_resample_i8d: the longest span that fits), one period tile per span with walks of three spans (many
    # workgroup edges, every span staged both ways) and the all-double kernel
    for tuned, dst in (({}, outs), ({"rs_i16_tiles": 1, "rs_i16_walk": 3}, outs_first), ({"rs_i16_path": 1}, outs_plain)):
        with capi.tuned(**tuned):
            r = filters.ResampleMC(ch, L, M, gain, win, filters.PCM_I16)
            for (o, e) in ((0, cut), (cut, nin)):
                xi = torch.from_numpy(np.ascontiguousarray(x[:, o:e])).to(dev)
                yi = torch.empty(ch, (e - o) // M * L, dtype=torch.int16, device=dev)
                assert r.process(xi, yi) == yi.shape[1]
                dst.append(yi.cpu().numpy())
            r.close()
    got, plain, first = np.concatenate(outs, axis=1), np.concatenate(outs_plain, axis=1), np.concatenate(outs_first, axis=1)
    assert np.array_equal(plain, ref), "all-double kernel vs oracle"
    for form in (got, first):
        bad = np.argwhere(form != ref)
        assert bad.size == 0, (L, M, len(bad), bad[:5].tolist(), form[tuple(bad[0])], ref[tuple(bad[0])])


@pytest.mark.parametrize("L,M", [(4, 3), (48, 7), (160, 147)])
def test_resample_i16_lm_exact_and_empty_phases(dev, L, M):
    """a caller's own polyphase matrix (llz_resample_mc_set_matrix) with a phase that is ONE tap 1.0 -- its outputs are the input
    samples themselves, integers the screen reproduces exactly and must not move toward zero -- and a phase with no tap at all
    (every output the integer 0): the screened kernel (two launch shapes) against the all-double kernel, which follows the
    reference's loop, and the exact phase against the samples it copies"""
    ch, periods = 5, 400
    rng = np.random.default_rng(L + M)
    x = rng.integers(-32768, 32767, (ch, M * periods), dtype=np.int64).astype(np.int16)
    x[1] = -32768
    x[2, ::2] = -1
    outs = {}
    for name, tuned in (("direct", {}), ("first", {"rs_i16_tiles": 2, "rs_i16_walk": 2}), ("double", {"rs_i16_path": 1})):
        with capi.tuned(**tuned):
            r = filters.ResampleMC(ch, L, M, 1.0, po.BLACKMAN, filters.PCM_I16)
            m = r.matrix()
            k = r.Q // 2
            m[0] = 0.0
            m[0, k] = 1.0
            m[1] = 0.0
            r.set_matrix(m)
            y = torch.empty(ch, L * periods, dtype=torch.int16, device=dev)
            assert r.process(torch.from_numpy(x).to(dev), y) == L * periods
            outs[name] = y.cpu().numpy()
            r.close()
    assert np.array_equal(outs["direct"], outs["double"]) and np.array_equal(outs["first"], outs["double"])
    # phase 0 of period m is x[M m - k] (zero history in front of the first call); phase 1 is silence
    got0 = outs["direct"][:, 0::L]
    want0 = np.zeros_like(got0)
    idx = M * np.arange(periods) - k
    want0[:, idx >= 0] = x[:, idx[idx >= 0]]
    assert np.array_equal(got0, want0)
    assert not outs["direct"][:, 1::L].any()


# ------------------------------------------------------------------------------------------------ general direct form I, many channels
@pytest.mark.parametrize("a,b", [
    ([1.0, -0.3695, 0.1958, 0.0], [1.0, 0.2066, 0.4131, 0.2066]),        # the reference's own order-3 call (llz_musicpitch.c:1277-1285)
    ([1.0, -1.2, 0.9, -0.35, 0.12, -0.02], [0.05, 0.1, 0.05]),            # M = 5, N = 2
    ([1.0], [0.25, 0.5, -0.125, 0.0625, 0.3]),                            # M = 0: feed-forward only
    ([1.0, -1.8 * 0.99 * 0.95, 0.99 * 0.99], [0.01]),                     # M = 2, N = 0, pole radius 0.99: long memory
])
def test_iir_mc_general_orders(dev, oracle, a, b):
    """llz_iir_mc_*: the reference's general direct-form-I filter (any orders) for 1024 channels at once, float32 in / out.
    (1) one time segment per channel: the reference's own double sequence rounded once to float32 -- equal to the oracle's
    llz_iir_filter bit for bit; (2) the default launch (channels split along time, later segments warmed up from zero delay
    lines): within 1e-5 RMS, streamed over two calls + flush."""
    channels, n = 1024, 24000
    x = oracle.synth_f32(channels, 2 * n, seed=len(a) * 10 + len(b))
    sel = [0, 1, 511, 1023]
    ref = np.stack([np.concatenate(oracle.iir_stream(np.array(a), np.array(b), x[c].astype(np.float64), flush=True)) for c in sel])
    xd = torch.from_numpy(x).to(dev)
    for segs in (1, -1):
        with capi.tuned(iir_segs=segs):
            f = filters.IirMC(channels, a, b)
            outs = []
            for o in (0, n):
                y = torch.empty(channels, n, dtype=torch.float32, device=dev)
                f.filter(xd[:, o:o + n].contiguous(), y)
                outs.append(y.cpu().numpy())
            N = len(b) - 1
            tail = np.zeros((channels, max(N, 1)), dtype=np.float32) if N == 0 else np.zeros((channels, N), dtype=np.float32)
            assert f.flush(tail) == N
            f.close()
        got = np.concatenate(outs + ([tail[:, :N]] if N else []), axis=1)[sel]
        if segs == 1:
            assert np.array_equal(got, ref.astype(np.float32)), (a, b)
        else:
            rms_check(got, ref, f"iir_mc M={len(a) - 1} N={len(b) - 1} (time segments)")
    with pytest.raises(capi.LlzError):
        filters.IirMC(4, [1.0] + [0.01] * 9, [1.0])                       # order 9: refused
