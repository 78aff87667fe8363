"""BASELINE config 2 (64 ch x 63 taps x 2^20) on its forms, with settled clocks: python tools/time_cfg2.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from llzlab_amd import capi, filters
dev = torch.device("cuda:0")
L = capi.lib()
ch, n = 64, 1 << 20
x = torch.empty(ch, n, dtype=torch.float32, device=dev); y = torch.empty_like(x)
filters.synth_f32(x, 1)
taps = filters.fir_design("lpf", 63, 0.25, 0.0, filters.HAMMING)
def timed(fn, reps=400):
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    t = L.llz_hip_timer_new(); L.llz_hip_timer_start(t, None)
    for _ in range(reps): fn()
    L.llz_hip_timer_stop(t, None); ms = L.llz_hip_timer_ms(t) / reps; L.llz_hip_timer_free(t)
    return ms
for name, algo, tune in (("time domain", 1, {}), ("overlap-save", 2, {}), ("overlap-save 3 wg/cu", 2, {"ols_wg_per_cu": 3})):
    with capi.tuned(**tune):
        f = filters.FirFilterMC(ch, n, taps, algo=algo)
        ms = timed(lambda: f.filter(x, y))
        print(f"{name:22s}: {ms:.4f} ms  {8 * ch * n / ms / 1e6:.0f} GB/s ({8 * ch * n / ms / 1e6 / 80:.1f} %)", flush=True)
        f.close()
