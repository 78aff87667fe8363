import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from llzlab_amd import capi, filters
dev = torch.device("cuda:0")
L = capi.lib()
def timed(fn, steps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t = L.llz_hip_timer_new(); L.llz_hip_timer_start(t, None)
    for _ in range(steps): fn()
    L.llz_hip_timer_stop(t, None); ms = L.llz_hip_timer_ms(t) / steps; L.llz_hip_timer_free(t)
    return ms
for ch, n in ((1024, 1 << 20), (64, 1 << 20), (8192, 1 << 16)):
    x = torch.rand(ch, n, dtype=torch.float32, device=dev) * 2 - 1
    y = torch.empty_like(x)
    for a, b in (([1.0, -0.3695, 0.1958, 0.0], [1.0, 0.2066, 0.4131, 0.2066]), ([1.0, -1.2, 0.9, -0.35, 0.12, -0.02], [0.05, 0.1, 0.05])):
        q = filters.IirMC(ch, a, b)
        ms = timed(lambda: q.filter(x, y))
        print(f"iir_mc M={len(a)-1} N={len(b)-1} {ch}ch x {n}: {ms:.3f} ms {8*ch*n/ms/1e6:.0f} GB/s ({8*ch*n/ms/1e6/80:.1f} %)", flush=True)
        q.close()
