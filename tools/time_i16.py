#!/usr/bin/env python3
"""Event-timed int16 1:3 decimator (LLZ_PCM_I16, the screened bit-exact path): python tools/time_i16.py [channels] [L M]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from llzlab_amd import capi, filters  # noqa: E402

ch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L_, M_ = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, 3)
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
Lb = capi.lib()
capi.check(Lb.llz_hip_set_device(0), "set_device")
stream = torch.cuda.current_stream()
n = M_ * (((1 << 22) // M_) // 256 * 256) if L_ == 1 else M_ * 8192
x = torch.empty(ch, n, dtype=torch.int16, device=dev)
y = torch.empty(ch, n * L_ // M_, dtype=torch.int16, device=dev)
filters.synth_i16(x, 1, stream=stream)
r = filters.ResampleMC(ch, L_, M_, 1.0, filters.BLACKMAN, filters.PCM_I16, stream=stream)
for _ in range(3 if ch >= 4096 else 60):           # (short kernels: tens of ms of work before the clocks have settled)
    r.process(x, y)
torch.cuda.synchronize()
t = Lb.llz_hip_timer_new()
reps = 5 if ch >= 4096 else 40
Lb.llz_hip_timer_start(t, stream.cuda_stream)
for _ in range(reps):
    r.process(x, y)
Lb.llz_hip_timer_stop(t, stream.cuda_stream)
ms = Lb.llz_hip_timer_ms(t) / reps
gb = (2 + 2 * L_ / M_) * ch * n / ms / 1e6
print(f"{os.path.basename(capi.LIB_PATH)}: resample {L_}:{M_} i16 exact {ch}ch x {n}: {ms:.3f} ms  {gb:.0f} GB/s ({gb / 80:.1f} %)")
if L_ > 1 and len(sys.argv) > 4:
    # sweep of the launch shape: period tiles per span x spans per workgroup
    for tiles in (1, 2, 3, 4):
        for walk in (2, 4, 8, 16, 32, 64, 128, 512):
            capi.tune("rs_i16_tiles", tiles)
            capi.tune("rs_i16_walk", walk)
            for _ in range(10):
                r.process(x, y)
            Lb.llz_hip_timer_start(t, stream.cuda_stream)
            for _ in range(20):
                r.process(x, y)
            Lb.llz_hip_timer_stop(t, stream.cuda_stream)
            print(f"   tiles {tiles} walk {walk}: {Lb.llz_hip_timer_ms(t) / 20:.3f} ms")
    capi.tune("rs_i16_tiles", -1)
    capi.tune("rs_i16_walk", -1)
if L_ > 1:
    import ctypes
    plan = (ctypes.c_int * 7)()
    if Lb.llzs_resample_i16x_plan(L_, M_, r.Q, ch, ctypes.c_long(n * L_ // M_), 38, plan) == 0:
        print("   plan: waves/wg %d, periods/span %d, spans/wg %d, workgroups %d, resident/CU %d, LDS %d B, several tiles per wave %d" % tuple(plan))
