#!/usr/bin/env python3
"""Sweep the IIR cascade's time-segment count (llz_hip_tune("iir_segs", n)) on config 4's shape: python tools/iir_segs.py [channels] [n]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from llzlab_amd import capi, filters  # noqa: E402

ch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
dev = torch.device("cuda:0")
L = capi.lib()
capi.check(L.llz_hip_set_device(0), "set_device")
stream = torch.cuda.current_stream()
sptr = stream.cuda_stream
x = torch.empty(ch, n, dtype=torch.float32, device=dev)
y = torch.empty_like(x)
filters.synth_f32(x, 1, stream=stream)
row = [0.2066, 0.4131, 0.2066, 1.0, -0.3695, 0.1958]
if os.environ.get("HIGHQ"):
    row = [0.01, 0.0, -0.01, 1.0, -2 * 0.99 * np.cos(0.3), 0.99 ** 2]
q = filters.IirCascadeMC(ch, np.tile(np.array(row), (8, 1)), stream=stream)
for segs in [0] + [int(v) for v in os.environ.get("SEGS", "3,6,8,9,12,18,24,48,64").split(",") if int(v) > 0]:
    capi.tune("iir_segs", segs if segs else -1)
    for _ in range(max(3, 40 * 1024 // ch)):                            # clocks settle over tens of milliseconds
        q.filter(x, y)
    torch.cuda.synchronize()
    t = L.llz_hip_timer_new()
    L.llz_hip_timer_start(t, sptr)
    reps = max(5, 20 * 1024 // ch)
    for _ in range(reps):
        q.filter(x, y)
    L.llz_hip_timer_stop(t, sptr)
    ms = L.llz_hip_timer_ms(t) / reps
    L.llz_hip_timer_free(t)
    print(f"{ch}ch x {n} segs={segs or 'auto'}: {ms:.3f} ms  {ch * n / ms / 1e3:.0f} Msamples/s ({8 * ch * n / ms / 1e6 / 80:.1f} %)", flush=True)
