"""headline FIR (4096 ch x 2^20 x 257 taps): jobs per segment of the walk (16 = default; 1 = every half-wave strides over the
whole batch job by job, i.e. the chip works on one compact window): python tools/time_ols_seg.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from llzlab_amd import capi, filters
dev = torch.device("cuda:0")
ch, n = 4096, 1 << 20
x = torch.empty(ch, n, dtype=torch.float32, device=dev)
y = torch.empty_like(x); y0 = torch.empty_like(x)
filters.synth_f32(x, 1)
L = capi.lib()
taps = filters.fir_design("lpf", 257, 0.1, 0.0, filters.KAISER)
def timed(fn, steps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t = L.llz_hip_timer_new(); L.llz_hip_timer_start(t, None)
    for _ in range(steps): fn()
    L.llz_hip_timer_stop(t, None); ms = L.llz_hip_timer_ms(t) / steps; L.llz_hip_timer_free(t)
    return ms
for sl in (16, 20, 24, 32, 43, 48, 64, 86, 16):
    with capi.tuned(ols_seg_len=sl):
        f = filters.FirFilterMC(ch, n, taps, algo=2)
        dst = y0 if sl == 16 else y
        ms = timed(lambda: f.filter(x, dst))
        print(f"seg_len {sl:2d}: {ms:.3f} ms  {8 * ch * n / ms / 1e6:.0f} GB/s ({8 * ch * n / ms / 1e6 / 80:.1f} %)  max|diff vs 16| {(dst - y0).abs().max().item():.3g}", flush=True)
        f.close()
