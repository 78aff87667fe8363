#!/usr/bin/env python3
"""Run one path a few times (for rocprofv3):  python3 tools/run_path.py {resample|fir_td|fir_ols|iir|resample_i16} [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from llzlab_amd import capi, filters  # noqa: E402

what = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
capi.check(capi.lib().llz_hip_set_device(0), "set_device")
if what == "resample":
    ch = 1024
    n = 3 * (((1 << 22) // 3) // 256 * 256)
    x = torch.empty(ch, n, dtype=torch.float32, device=dev)
    y = torch.empty(ch, n // 3, dtype=torch.float32, device=dev)
    filters.synth_f32(x, 1)
    r = filters.ResampleMC(ch, 1, 3, 1.0, filters.BLACKMAN, filters.PCM_F32)
    for _ in range(steps):
        r.process(x, y)
elif what == "resample_i16":
    ch = 1024
    n = 3 * (((1 << 22) // 3) // 256 * 256)
    x = torch.empty(ch, n, dtype=torch.int16, device=dev)
    y = torch.empty(ch, n // 3, dtype=torch.int16, device=dev)
    filters.synth_i16(x, 1)
    r = filters.ResampleMC(ch, 1, 3, 1.0, filters.BLACKMAN, filters.PCM_I16)
    for _ in range(steps):
        r.process(x, y)
elif what in ("rs147", "rs160"):
    L_, M_ = (147, 160) if what == "rs147" else (160, 147)
    ch, n = 256, M_ * 8192
    x = torch.empty(ch, n, dtype=torch.float32, device=dev)
    y = torch.empty(ch, n * L_ // M_, dtype=torch.float32, device=dev)
    filters.synth_f32(x, 1)
    r = filters.ResampleMC(ch, L_, M_, 1.0, filters.BLACKMAN, filters.PCM_F32)
    for _ in range(steps * 5):
        r.process(x, y)
elif what in ("fir_td", "fir_ols"):
    ch, n = 4096, 1 << 20
    x = torch.empty(ch, n, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    filters.synth_f32(x, 1)
    f = filters.FirFilterMC(ch, n, filters.fir_design("lpf", 257, 0.1, 0.0, filters.KAISER),
                            algo=1 if what == "fir_td" else 2)
    for _ in range(steps):
        f.filter(x, y)
elif what == "fir63":
    ch, n = 4096, 1 << 20
    x = torch.empty(ch, n, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    filters.synth_f32(x, 1)
    f = filters.FirFilterMC(ch, n, filters.fir_design("lpf", 63, 0.25, 0.0, filters.HAMMING), algo=1)
    for _ in range(steps):
        f.filter(x, y)
elif what == "fft":
    N, count = 1024, 65536
    data = torch.rand(count, 2 * N, dtype=torch.float32, device=dev)
    f = filters.FftBatch(N)
    for _ in range(steps):
        f.fft(data, count)
elif what == "iir":
    ch, n = 1024, 1 << 20
    x = torch.empty(ch, n, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    filters.synth_f32(x, 1)
    coef = np.tile(np.array([0.2066, 0.4131, 0.2066, 1.0, -0.3695, 0.1958]), (8, 1))
    q = filters.IirCascadeMC(ch, coef)
    for _ in range(steps):
        q.filter(x, y)
elif what == "traffic_set":
    # every kernel whose HBM traffic profiles/pmc_traffic.json records, a few launches each, one process (tools/profile_round.sh)
    def go(fn, k=steps):
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
    ch, n = 1024, 1 << 20                                           # BASELINE config 4, both coefficient sets
    x = torch.empty(ch, n, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    filters.synth_f32(x, 1)
    for row in ([0.2066, 0.4131, 0.2066, 1.0, -0.3695, 0.1958], [0.01, 0.0, -0.01, 1.0, -2 * 0.99 * np.cos(0.3), 0.99 ** 2]):
        q = filters.IirCascadeMC(ch, np.tile(np.array(row), (8, 1)))
        go(lambda: q.filter(x, y))
        q.close()
    del x, y
    ch = 2048                                                       # BASELINE config 5 (a quarter of its channels), three formats
    n = 3 * (((1 << 22) // 3) // 256 * 256)
    x = torch.empty(ch, n, dtype=torch.float32, device=dev)
    y = torch.empty(ch, n // 3, dtype=torch.float32, device=dev)
    filters.synth_f32(x, 1)
    r = filters.ResampleMC(ch, 1, 3, 1.0, filters.BLACKMAN, filters.PCM_F32)
    go(lambda: r.process(x, y))
    r.close()
    del x, y
    xi = torch.empty(ch, n, dtype=torch.int16, device=dev)
    yi = torch.empty(ch, n // 3, dtype=torch.int16, device=dev)
    filters.synth_i16(xi, 1)
    for fmt in (filters.PCM_I16, filters.PCM_I16_FAST):
        # (a FAST handle would take the screened kernel too: its own float32-sum kernel is what is recorded here)
        with capi.tuned(rs_i16_path=1 if fmt == filters.PCM_I16_FAST else -1):
            r = filters.ResampleMC(ch, 1, 3, 1.0, filters.BLACKMAN, fmt)
        go(lambda: r.process(xi, yi))
        r.close()
    del xi, yi
    for (L_, M_) in ((147, 160), (160, 147)):                       # the reference CLI's ratios, float32 and int16
        ch, n = 256, M_ * 8192
        x = torch.empty(ch, n, dtype=torch.float32, device=dev)
        y = torch.empty(ch, n * L_ // M_, dtype=torch.float32, device=dev)
        filters.synth_f32(x, 1)
        r = filters.ResampleMC(ch, L_, M_, 1.0, filters.BLACKMAN, filters.PCM_F32)
        go(lambda: r.process(x, y))
        r.close()
        xi = torch.empty(ch, n, dtype=torch.int16, device=dev)
        yi = torch.empty(ch, n * L_ // M_, dtype=torch.int16, device=dev)
        filters.synth_i16(xi, 1)
        r = filters.ResampleMC(ch, L_, M_, 1.0, filters.BLACKMAN, filters.PCM_I16)
        go(lambda: r.process(xi, yi))
        r.close()
        del x, y, xi, yi
    # rank 4 of the widening: fixed-point MDCT (N/4-point form, one launch), MDCT-frames synthesis in runs of segments
    n, count = 2048, 1 << 16
    xq = torch.randint(-(1 << 20), 1 << 20, (count, n), dtype=torch.int32, device=dev)
    Xq = torch.empty(count, n // 2, dtype=torch.int32, device=dev)
    mq = filters.MdctFixed(2, n)
    go(lambda: mq.forward_batch(xq, Xq))
    go(lambda: mq.inverse_batch(Xq, xq))
    mq.close()
    del xq, Xq
    F, ch, frames = 256, 1024, 1024
    xf = torch.rand(ch, frames * F, dtype=torch.float32, device=dev) * 2 - 1
    Xf = torch.empty(ch, frames, F, dtype=torch.float32, device=dev)
    mf = filters.MdctFramesMC(ch, F, capi.MDCT_SINE)
    go(lambda: mf.analysis(xf, Xf))
    go(lambda: mf.synthesis(Xf, xf))
    mf.close()
    del xf, Xf
    ch, n = 2048, 1 << 20                                           # long FIR: 3073 taps on the 8192-point pairs-of-waves kernel
    x = torch.empty(ch, n, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    filters.synth_f32(x, 1)
    lf = filters.FirFilterMC(ch, n, filters.fir_design("lpf", 3073, 0.1, 0.0, filters.KAISER))
    go(lambda: lf.filter(x, y))
    lf.close()
    del x, y
torch.cuda.synchronize()
print("done", what)
