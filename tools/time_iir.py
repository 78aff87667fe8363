"""IIR cascade on BASELINE config 4 (1024 ch x 2^20, 8 sections): python tools/time_iir.py [radius ...]   (0.44 = the config; 0.99 = the double set)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from llzlab_amd import capi, filters
radii = [float(v) for v in sys.argv[1:]] or [0.44, 0.99]
dev = torch.device("cuda:0")
ch, n = 1024, 1 << 20
x = torch.empty(ch, n, dtype=torch.float32, device=dev)
y = torch.empty_like(x)
filters.synth_f32(x, 1)
L = capi.lib()
def timed(fn, steps=40):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t = L.llz_hip_timer_new(); L.llz_hip_timer_start(t, None)
    for _ in range(steps): fn()
    L.llz_hip_timer_stop(t, None); ms = L.llz_hip_timer_ms(t) / steps; L.llz_hip_timer_free(t)
    return ms
for r in radii:
    a1, a2 = -2 * r * np.cos(0.3), r * r
    g = (1 + a1 + a2) / 4
    coef = np.tile(np.array([g, 2 * g, g, 1.0, a1, a2]), (8, 1))
    for name, tune in (("32 per lane", {}), ("16 per lane", {"iir_unpacked": 2}), ("32 per lane", {})):
        with capi.tuned(**tune):
            f = filters.IirCascadeMC(ch, coef)
            ms = timed(lambda: f.filter(x, y))
            print(f"iir 8 sections radius {r} f{f.precision} {name}: {ms:.3f} ms  {8 * ch * n / ms / 1e6:.0f} GB/s ({8 * ch * n / ms / 1e6 / 80:.1f} %)", flush=True)
            f.close()
