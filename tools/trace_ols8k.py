#!/usr/bin/env python3
"""Phase times of k_fir_ols8k_f32 (8192-point overlap-save on pairs of waves) from a -DO8K_TRACE build
(LLZ_LIB=llzlab_amd/libllz_var_o8ktrace.so): python tools/trace_ols8k.py [taps]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from llzlab_amd import capi, filters  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3073
dev = torch.device("cuda:0")
Lb = capi.lib()
ch, n = 4096, 1 << 20
x = torch.empty(ch, n, dtype=torch.float32, device=dev)
y = torch.empty_like(x)
filters.synth_f32(x, 1)
f = filters.FirFilterMC(ch, n, filters.fir_design("lpf", T, 0.1, 0.0, filters.KAISER), algo=filters.FIR_ALGO_OVERLAP_SAVE_8192)
for _ in range(3):
    f.filter(x, y)
torch.cuda.synchronize()
count = 2048
buf = np.zeros(count * 8, dtype=np.uint64)
Lb.llzs_o8k_trace_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(count * 8))
t = buf.reshape(count, 8).astype(np.float64)
role = (np.arange(count) % 8) >> 2            # the pair = waves p and p + 4 of a workgroup: wave 0 of the pair is p
role = role[t[:, 7] > 0]
t = t[t[:, 7] > 0]
jobs = t[:, 7]
names = ("requests awaited", "step down", "swap", "4096-point problem", "swap + turn", "outputs, stores issued", "loop")
print(f"{T} taps: {len(t)} waves, {jobs.mean():.1f} jobs each; shader clocks (s_memtime) per job")
for par, nm in ((0, "waves 0 of the pairs"), (1, "waves 1 of the pairs")):
    sel = t[role == par]
    print(f" {nm}")
    for i, name in enumerate(names):
        print(f"  {name:28s} {(sel[:, i] / sel[:, 7]).mean():9.0f}")
    print(f"  total {(sel[:, :7].sum(axis=1) / sel[:, 7]).mean():9.0f}")
