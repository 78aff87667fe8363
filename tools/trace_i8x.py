#!/usr/bin/env python3
"""Phase times of k_fir_mfma_i8x (the bit-exact int16 1:3 decimator) from a -DMX_TRACE build
(LLZ_LIB=llzlab_amd/libllzfilter_hip_trace.so): python tools/trace_i8x.py [channels]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from llzlab_amd import capi, filters  # noqa: E402

ch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
Lb = capi.lib()
capi.check(Lb.llz_hip_set_device(0), "set_device")
stream = torch.cuda.current_stream()
n = 3 * (((1 << 22) // 3) // 256 * 256)
x = torch.empty(ch, n, dtype=torch.int16, device=dev)
y = torch.empty(ch, n // 3, dtype=torch.int16, device=dev)
filters.synth_i16(x, 1, stream=stream)
r = filters.ResampleMC(ch, 1, 3, 1.0, filters.BLACKMAN, filters.PCM_I16, stream=stream)
for _ in range(4):
    r.process(x, y)
torch.cuda.synchronize()
count = 65536
buf = np.zeros(count * 8, dtype=np.uint64)
Lb.llzs_mx_trace_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(count * 8))
t = buf.reshape(count, 8).astype(np.float64)
t = t[t[:, 7] > 0]
tiles = t[:, 7]
names = ("first barrier", "request awaited, planes written", "second barrier", "next tile requested", "operand reads, products",
         "decisions", "stores")
print(f"{len(t)} waves, {tiles.mean():.1f} tiles each; shader clocks per tile")
for i, nm in enumerate(names):
    print(f"  {nm:34s} {(t[:, i] / tiles).mean():9.0f}")
print(f"  total {(t[:, :7].sum(axis=1) / tiles).mean():9.0f}")
