"""general L/M float32 resampler, matrix-core form against the LDS form: python tools/time_rs_general.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from llzlab_amd import capi, filters
dev = torch.device("cuda:0")
L = capi.lib()
def timed(fn, steps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t = L.llz_hip_timer_new(); L.llz_hip_timer_start(t, None)
    for _ in range(steps): fn()
    L.llz_hip_timer_stop(t, None); ms = L.llz_hip_timer_ms(t) / steps; L.llz_hip_timer_free(t)
    return ms
for (L_, M_) in ((147, 160), (160, 147), (8, 7), (5, 6), (441, 320), (320, 441), (48, 125)):
    ch = 256
    n = M_ * (8192 if M_ > 100 else 200000)
    x = torch.empty(ch, n, dtype=torch.float32, device=dev)
    y = torch.empty(ch, n * L_ // M_, dtype=torch.float32, device=dev)
    y2 = torch.empty_like(y)
    filters.synth_f32(x, 1)
    out = {}
    arms = (("phase-tile waves", {}, y), ("period-tile waves", {"rs_mfma_form": 1}, y2), ("LDS form", {"rs_generic": 1}, y2))
    for name, tune, dst in arms:
        with capi.tuned(**tune):
            r = filters.ResampleMC(ch, L_, M_, 1.0, filters.BLACKMAN, filters.PCM_F32)
            ms = timed(lambda: r.process(x, dst))
            gb = (4 + 4 * L_ / M_) * ch * n / ms / 1e6
            print(f"resample {L_}:{M_} Q={r.Q} {name:17s} {ch}ch x {n}: {ms:.3f} ms  {gb:.0f} GB/s ({gb / 80:.1f} % of 8 TB/s)", flush=True)
            r.close()
    # both handles ran the same frames the same number of times: same history
    err = (y - y2).abs().max().item()
    print(f"   max |matrix - LDS| = {err:.3g}")
    del x, y, y2
