#!/usr/bin/env python3
"""How many workgroups of the bit-exact int16 L/M resampler a CU really runs at a time: every workgroup gets the same fixed walk
(16 spans of 48 periods), the channel count sets the number of workgroups to about k x 256.  Time flat in k up to the residency,
then in steps.  python tools/time_i16_residency.py [L M]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from llzlab_amd import capi, filters  # noqa: E402

L_, M_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (147, 160)
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
Lb = capi.lib()
capi.check(Lb.llz_hip_set_device(0), "set_device")
stream = torch.cuda.current_stream()
n = M_ * 8192
capi.tune("rs_i16_tiles", 3)
capi.tune("rs_i16_walk", 16)
per_ch = -(-(-(-8192 // 48)) // 16)          # workgroups per channel: ceil(ceil(8192 / 48) / 16)
t = Lb.llz_hip_timer_new()
for k in (0.25, 0.5, 1, 1.5, 2, 2.5, 3, 4, 5, 6, 8):
    ch = max(1, round(256 * k / per_ch))
    x = torch.empty(ch, n, dtype=torch.int16, device=dev)
    y = torch.empty(ch, n * L_ // M_, dtype=torch.int16, device=dev)
    filters.synth_i16(x, 1, stream=stream)
    r = filters.ResampleMC(ch, L_, M_, 1.0, filters.BLACKMAN, filters.PCM_I16, stream=stream)
    for _ in range(30):
        r.process(x, y)
    Lb.llz_hip_timer_start(t, stream.cuda_stream)
    for _ in range(30):
        r.process(x, y)
    Lb.llz_hip_timer_stop(t, stream.cuda_stream)
    print(f"{ch:4d} channels = {ch * per_ch:5d} workgroups ({ch * per_ch / 256:.2f} per CU): {Lb.llz_hip_timer_ms(t) / 30:.3f} ms")
    r.close()
