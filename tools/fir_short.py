#!/usr/bin/env python3
"""Short filters: time-domain (algo 1) against overlap-save (algo 2) and the automatic choice (algo 0) at 4..64 taps."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from llzlab_amd import capi, filters  # noqa: E402

dev = torch.device("cuda:0")
L = capi.lib()
capi.check(L.llz_hip_set_device(0), "dev")
s = torch.cuda.current_stream()
sp = s.cuda_stream
ch, n = 4096, 1 << 20
x = torch.empty(ch, n, dtype=torch.float32, device=dev)
y = torch.empty_like(x)
filters.synth_f32(x, 1, stream=s)
for taps_n in (4, 8, 16, 24, 32, 33, 48, 64):
    taps = filters.fir_design("lpf", taps_n, 0.25, 0.0, filters.HAMMING)
    for algo in (0, 1, 2):
        f = filters.FirFilterMC(ch, n, taps, algo=algo, stream=s)
        for _ in range(3):
            f.filter(x, y)
        torch.cuda.synchronize()
        t = L.llz_hip_timer_new()
        L.llz_hip_timer_start(t, sp)
        for _ in range(5):
            f.filter(x, y)
        L.llz_hip_timer_stop(t, sp)
        ms = L.llz_hip_timer_ms(t) / 5
        L.llz_hip_timer_free(t)
        print(f"{ch}ch taps={taps_n} algo={algo}: {ms:.3f} ms {8 * ch * n / ms / 1e6:.0f} GB/s", flush=True)
        f.close()
