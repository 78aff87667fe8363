#!/usr/bin/env python3
"""Event-timed secondary paths on one GPU: python tools/time_paths.py [iir] [resample] [fir63] [td257]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from llzlab_amd import capi, filters  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
L = capi.lib()
capi.check(L.llz_hip_set_device(0), "set_device")
stream = torch.cuda.current_stream()
sptr = stream.cuda_stream


def timeit(fn, steps):
    """mean over >= `steps` calls and >= 30 ms, after >= 30 ms of warm-up calls (a sub-millisecond kernel timed right after
    idling runs on clocks that have not settled: up to 20 % slow)"""
    def run(k):
        t = L.llz_hip_timer_new()
        L.llz_hip_timer_start(t, sptr)
        for _ in range(k):
            fn()
        L.llz_hip_timer_stop(t, sptr)
        ms = L.llz_hip_timer_ms(t) / k
        L.llz_hip_timer_free(t)
        return ms
    fn()
    torch.cuda.synchronize()
    once = max(run(1), 1e-3)
    reps = max(steps, min(2000, int(30.0 / once) + 1))
    run(reps)
    return run(reps)


which = sys.argv[1:] or ["iir", "resample", "fir63", "td257", "fft", "corr", "pcm", "stft", "mdct"]
if "iir" in which:
    for ch in (1024, 128):
        n = 1 << 20
        x = torch.empty(ch, n, dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        filters.synth_f32(x, 1, stream=stream)
        for name, row in (("r0.44", [0.2066, 0.4131, 0.2066, 1.0, -0.3695, 0.1958]),
                          ("r0.99", [0.01, 0.0, -0.01, 1.0, -2 * 0.99 * np.cos(0.3), 0.99 ** 2])):
            q = filters.IirCascadeMC(ch, np.tile(np.array(row), (8, 1)), stream=stream)
            ms = timeit(lambda: q.filter(x, y), 5)
            print(f"iir8 {name} {ch}ch x {n}: {ms:.3f} ms  {ch * n / ms / 1e3:.0f} Msamples/s  {8 * ch * n / ms / 1e6:.0f} GB/s "
                  f"({8 * ch * n / ms / 1e6 / 80:.1f} % of 8 TB/s)")
            q.close()
        del x, y
if "resample" in which:
    for ch in (8192, 1024):
        n = 3 * (((1 << 22) // 3) // 256 * 256)
        x = torch.empty(ch, n, dtype=torch.float32, device=dev)
        y = torch.empty(ch, n // 3, dtype=torch.float32, device=dev)
        filters.synth_f32(x, 1, stream=stream)
        r = filters.ResampleMC(ch, 1, 3, 1.0, filters.BLACKMAN, filters.PCM_F32, stream=stream)
        ms = timeit(lambda: r.process(x, y), 3)
        gb = (4 + 4 / 3) * ch * n / ms / 1e6
        print(f"resample 1:3 f32 {ch}ch x {n}: {ms:.3f} ms  {ch * n / ms / 1e3:.0f} Msamples_in/s  {gb:.0f} GB/s ({gb / 80:.1f} %)")
        r.close()
        del x, y
if "resample_i16" in which:
    ch = 1024
    n = 3 * (((1 << 22) // 3) // 256 * 256)
    x = torch.empty(ch, n, dtype=torch.int16, device=dev)
    y = torch.empty(ch, n // 3, dtype=torch.int16, device=dev)
    filters.synth_i16(x, 1, stream=stream)
    r = filters.ResampleMC(ch, 1, 3, 1.0, filters.BLACKMAN, filters.PCM_I16, stream=stream)
    ms = timeit(lambda: r.process(x, y), 2)
    gb = (2 + 2 / 3) * ch * n / ms / 1e6
    print(f"resample 1:3 i16 exact {ch}ch x {n}: {ms:.3f} ms  {ch * n / ms / 1e3:.0f} Msamples_in/s  {gb:.0f} GB/s ({gb / 80:.1f} %)")
    r.close()
    y.zero_()
    r = filters.ResampleMC(ch, 1, 3, 1.0, filters.BLACKMAN, filters.PCM_I16_FAST, stream=stream)
    ms = timeit(lambda: r.process(x, y), 3)
    gb = (2 + 2 / 3) * ch * n / ms / 1e6
    print(f"resample 1:3 i16 fast  {ch}ch x {n}: {ms:.3f} ms  {ch * n / ms / 1e3:.0f} Msamples_in/s  {gb:.0f} GB/s ({gb / 80:.1f} %)")
    r.close()
    for (L_, M_) in ((2, 3), (3, 2), (147, 160), (160, 147)):
        ch = 256
        n = M_ * (8192 if M_ > 100 else 400000)
        x = torch.empty(ch, n, dtype=torch.float32, device=dev)
        y = torch.empty(ch, n * L_ // M_, dtype=torch.float32, device=dev)
        filters.synth_f32(x, 1, stream=stream)
        r = filters.ResampleMC(ch, L_, M_, 1.0, filters.BLACKMAN, filters.PCM_F32, stream=stream)
        ms = timeit(lambda: r.process(x, y), 3)
        gb = (4 + 4 * L_ / M_) * ch * n / ms / 1e6
        print(f"resample {L_}:{M_} f32 generic {ch}ch x {n}: {ms:.3f} ms  {ch * n / ms / 1e3:.0f} Msamples_in/s  {gb:.0f} GB/s ({gb / 80:.1f} %)")
        r.close()
        del x, y
if "fir63" in which:
    for ch in (64, 4096):
        n = 1 << 20
        x = torch.empty(ch, n, dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        filters.synth_f32(x, 1, stream=stream)
        f = filters.FirFilterMC(ch, n, filters.fir_design("lpf", 63, 0.25, 0.0, filters.HAMMING), algo=1, stream=stream)
        ms = timeit(lambda: f.filter(x, y), 10)
        print(f"fir63 td {ch}ch x {n}: {ms:.3f} ms  {ch * n / ms / 1e3:.0f} Msamples/s  {8 * ch * n / ms / 1e6:.0f} GB/s "
              f"({8 * ch * n / ms / 1e6 / 80:.1f} %)")
        f.close()
        del x, y
if "td257" in which:
    ch, n = 4096, 1 << 20
    x = torch.empty(ch, n, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    filters.synth_f32(x, 1, stream=stream)
    f = filters.FirFilterMC(ch, n, filters.fir_design("lpf", 257, 0.1, 0.0, filters.KAISER), algo=1, stream=stream)
    ms = timeit(lambda: f.filter(x, y), 3)
    print(f"fir257 td {ch}ch x {n}: {ms:.3f} ms  {8 * ch * n / ms / 1e6:.0f} GB/s")
    f.close()

if "fft" in which:
    for n in (64, 128, 256, 512, 1024, 2048, 4096):
        count = (1 << 26) // n                                   # 512 MiB of complex64 / int32 pairs
        z = torch.rand(count, 2 * n, dtype=torch.float32, device=dev) * 2 - 1
        f = filters.FftBatch(n, stream=stream)
        ms = timeit(lambda: f.fft(z, count), 3)
        print(f"fft f32 N={n} x {count}: {ms:.3f} ms  {16 * n * count / ms / 1e6:.0f} GB/s ({16 * n * count / ms / 1e6 / 80:.1f} %)")
        f.close()
        q = torch.empty(count, 2 * n, dtype=torch.int32, device=dev)
        q.copy_((z * 1000).to(torch.int32))
        fx = filters.FftFixed(n, stream=stream)
        ms = timeit(lambda: fx.fft_batch(q, count), 3)
        print(f"fft q15 N={n} x {count}: {ms:.3f} ms  {16 * n * count / ms / 1e6:.0f} GB/s ({16 * n * count / ms / 1e6 / 80:.1f} %)")
        fx.close()
        del z, q

if "corr" in which:
    for (frames, n, p) in ((1 << 18, 1024, 16), (1 << 18, 1024, 64), (1 << 19, 512, 16), (1 << 20, 256, 16), (1 << 17, 2048, 16),
                           (1 << 17, 2048, 255), (4096, 1 << 18, 32)):
        x = torch.rand(frames, n, dtype=torch.float32, device=dev) * 2 - 1
        r = torch.empty(frames, p + 1, dtype=torch.float32, device=dev)
        ms = timeit(lambda: filters.autocorr_mc(x, r, p, stream=stream), 3)
        gb = 4.0 * frames * n / ms / 1e6
        print(f"autocorr direct {frames} x {n}, p={p}: {ms:.3f} ms  {frames * n / ms / 1e3:.0f} Msamples/s  {gb:.0f} GB/s ({gb / 80:.1f} %)")
        if n <= 2048:
            f = filters.AutocorrFastMC(frames, n, stream=stream)
            ms = timeit(lambda: f.run(x, r, p), 3)
            gb = 4.0 * frames * n / ms / 1e6
            print(f"autocorr fft    {frames} x {n}, p={p}: {ms:.3f} ms  {frames * n / ms / 1e3:.0f} Msamples/s  {gb:.0f} GB/s ({gb / 80:.1f} %)")
            if n == 2048:
                with capi.tuned(fft_generic=1):
                    ms = timeit(lambda: f.run(x, r, p), 3)
                print(f"autocorr fft    {frames} x {n}, p={p} (staged passes): {ms:.3f} ms")
            f.close()
        del x, r

if "pcm" in which:
    for ch in (2, 64, 4096):
        n = (1 << 31) // ch // 2                                          # 2 GiB of int16 in, 4 GiB of float32 out
        il = torch.randint(-32768, 32767, (n, ch), dtype=torch.int16, device=dev)
        pl = torch.empty(ch, n, dtype=torch.float32, device=dev)
        ms = timeit(lambda: filters.pcm_deinterleave(il, pl, stream=stream), 3)
        gb = 6.0 * ch * n / ms / 1e6
        print(f"pcm deinterleave {ch}ch x {n}: {ms:.3f} ms  {gb:.0f} GB/s ({gb / 80:.1f} %)")
        ms = timeit(lambda: filters.pcm_interleave(pl, il, stream=stream), 3)
        gb = 6.0 * ch * n / ms / 1e6
        print(f"pcm interleave   {ch}ch x {n}: {ms:.3f} ms  {gb:.0f} GB/s ({gb / 80:.1f} %)")
        del il, pl

if "stft" in which:
    for hint, F, ch, frames in ((0, 256, 1024, 256), (1, 512, 1024, 128), (0, 64, 4096, 256), (0, 512, 1024, 128),
                                (0, 256, 8, 32768)):
        n = F * frames
        x = torch.rand(ch, n, dtype=torch.float32, device=dev) * 2 - 1
        q = filters.StftMC(ch, hint, F, filters.BLACKMAN, stream=stream)
        re = torch.empty(ch, frames, q.bins, dtype=torch.float32, device=dev)
        im = torch.empty_like(re)
        y = torch.empty_like(x)
        ms_a = timeit(lambda: q.analysis(x, re, im), 5)
        ms_s = timeit(lambda: q.synthesis(re, im, y), 5)
        spec = 8 * ch * frames * q.bins
        print(f"stft {'3/4' if hint == 0 else '1/2'} overlap F={F} {ch}ch x {frames} frames: analysis {ms_a:.3f} ms "
              f"{(4 * ch * n + spec) / ms_a / 1e6:.0f} GB/s, synthesis {ms_s:.3f} ms {(4 * ch * n + spec) / ms_s / 1e6:.0f} GB/s")
        q.close()
        del x, y, re, im

if "mdct" in which:
    for n, count in ((256, 1 << 20), (2048, 1 << 17), (8192, 1 << 15)):
        x = torch.rand(count, n, dtype=torch.float32, device=dev) * 2 - 1
        X = torch.empty(count, n // 2, dtype=torch.float32, device=dev)
        q = filters.MdctBatch(n, stream=stream)
        ms_f = timeit(lambda: q.forward(x, X), 5)
        ms_i = timeit(lambda: q.inverse(X, x), 5)
        gb = 6 * count * n
        print(f"mdct N={n} x {count}: forward {ms_f:.3f} ms {gb / ms_f / 1e6:.0f} GB/s ({gb / ms_f / 1e6 / 80:.1f} %), "
              f"inverse {ms_i:.3f} ms {gb / ms_i / 1e6:.0f} GB/s")
        q.close()
        del x, X
    # windowed 50 %-overlap frames in batch: 4 B in + 4 B out per sample either way
    for F, ch, frames in ((128, 1024, 2048), (256, 1024, 1024), (512, 1024, 512), (1024, 1024, 256), (4096, 256, 256)):
        x = torch.rand(ch, frames * F, dtype=torch.float32, device=dev) * 2 - 1
        X = torch.empty(ch, frames, F, dtype=torch.float32, device=dev)
        q = filters.MdctFramesMC(ch, F, capi.MDCT_SINE, stream=stream)
        ms_a = timeit(lambda: q.analysis(x, X), 5)
        ms_s = timeit(lambda: q.synthesis(X, x), 5)
        gb = 8 * ch * frames * F
        print(f"mdct frames F={F} {ch}ch x {frames}: analysis {ms_a:.3f} ms {gb / ms_a / 1e6:.0f} GB/s ({gb / ms_a / 1e6 / 80:.1f} %), "
              f"synthesis {ms_s:.3f} ms {gb / ms_s / 1e6:.0f} GB/s ({gb / ms_s / 1e6 / 80:.1f} %)")
        if F <= 512:
            # the synthesis with a group per run of segments (the library's own choice here) against the two launches
            for forced in (0, 4, 16):
                capi.tune("mdct_run", forced)
                ms = timeit(lambda: q.synthesis(X, x), 5)
                print(f"   synthesis, mdct_run = {forced}: {ms:.3f} ms ({gb / ms / 1e6 / 80:.1f} %)")
            capi.tune("mdct_run", -1)
        q.close()
        del x, X

if "mdctq" in which:
    # fixed-point MDCT batch (int32 data, Q15 tables, bit-exact): 4 B per sample in + 2 B per sample out forward, the reverse inverse
    for t, n, count in ((2, 256, 1 << 20), (2, 2048, 1 << 17), (2, 8192, 1 << 15), (1, 2048, 1 << 15)):
        x = torch.randint(-(1 << 20), 1 << 20, (count, n), dtype=torch.int32, device=dev)
        X = torch.empty(count, n // 2, dtype=torch.int32, device=dev)
        q = filters.MdctFixed(t, n)
        q.set_stream(stream)
        ms_f = timeit(lambda: q.forward_batch(x, X), 5)
        ms_i = timeit(lambda: q.inverse_batch(X, x), 5)
        gb = 6 * count * n
        print(f"mdct fixed type {t} N={n} x {count}: forward {ms_f:.3f} ms {gb / ms_f / 1e6:.0f} GB/s ({gb / ms_f / 1e6 / 80:.1f} %), "
              f"inverse {ms_i:.3f} ms {gb / ms_i / 1e6:.0f} GB/s ({gb / ms_i / 1e6 / 80:.1f} %)")
        if t in (1, 2):
            with capi.tuned(mdctq_steps=1):
                ms_f = timeit(lambda: q.forward_batch(x, X), 5)
                ms_i = timeit(lambda: q.inverse_batch(X, x), 5)
            print(f"   as three launches (step, transform, step): forward {ms_f:.3f} ms ({gb / ms_f / 1e6 / 80:.1f} %), "
                  f"inverse {ms_i:.3f} ms ({gb / ms_i / 1e6 / 80:.1f} %)")
        q.close()
        del x, X
