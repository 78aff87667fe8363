#!/usr/bin/env python3
"""Phase times of k_resample_i8x from a -DRI_TRACE build (LLZ_LIB=llzlab_amd/libllzfilter_hip_trace.so): python tools/trace_i16.py [ch L M]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from llzlab_amd import capi, filters  # noqa: E402

ch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L_, M_ = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (147, 160)
dev = torch.device("cuda:0")
Lb = capi.lib()
capi.check(Lb.llz_hip_set_device(0), "set_device")
stream = torch.cuda.current_stream()
n = M_ * 8192
x = torch.empty(ch, n, dtype=torch.int16, device=dev)
y = torch.empty(ch, n * L_ // M_, dtype=torch.int16, device=dev)
filters.synth_i16(x, 1, stream=stream)
r = filters.ResampleMC(ch, L_, M_, 1.0, filters.BLACKMAN, filters.PCM_I16, stream=stream)
for _ in range(20):
    r.process(x, y)
torch.cuda.synchronize()
plan = (ctypes.c_int * 7)()
Lb.llzs_resample_i16x_plan(L_, M_, r.Q, ch, ctypes.c_long(n * L_ // M_), 38, plan)
waves, wgs = plan[0], plan[3]
count = min(65536, waves * wgs)
buf = np.zeros(count * 10, dtype=np.uint64)
Lb.llzs_ri_trace_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(count * 10))
t = buf.reshape(count, 10).astype(np.float64)
names = ("planes written", "next span requested", "barrier (first form: staging barrier)", "(event counts)", "second barrier", "output stored",
         "  per-tile constants / rest of the tile loop", "  operand reads, products", "  decisions", "  second looks, image / stores")
print("plan: waves/wg %d, periods/span %d, spans/wg %d, workgroups %d, resident/CU %d, LDS %d B, several tiles per wave %d" % tuple(plan))
spans = plan[2]
print(f"shader-clock ticks per span, mean over {count} waves (and of wave 0 / the last wave of each workgroup)")
for i, nm in enumerate(names):
    col = t[:, i] / spans
    print(f"  {nm:46s} {col.mean():9.0f}   first wave {col[0::waves].mean():9.0f}   last wave {col[waves - 1::waves].mean():9.0f}")
t[:, 3] = 0
print(f"  total {t.sum(axis=1).mean() / spans:9.0f}")
ev = buf.reshape(count, 10)[:, 3]
taken, looks = (ev >> np.uint64(32)).astype(np.float64), (ev & np.uint64(0xffffffff)).astype(np.float64)
iters = spans * (plan[1] // 16)
print(f"tile iterations with an undecided output: {taken.mean() / iters * 100:.1f} % (first wave {taken[0::waves].mean() / iters * 100:.1f} %, "
      f"last wave {taken[waves - 1::waves].mean() / iters * 100:.1f} %); second looks per tile iteration {looks.mean() / iters:.3f}")
print("per wave of the workgroup (ticks per span):  " + "  ".join(n[:14].strip() for n in names if "event" not in n))
for w in range(waves):
    print(f"  wave {w:2d}: " + "  ".join(f"{t[w::waves, i].mean() / spans:8.0f}" for i in range(10) if i != 3))
