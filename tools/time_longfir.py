"""long FIR on the headline batch: python tools/time_longfir.py [taps[:algo] ...]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from llzlab_amd import capi, filters
cases = [v.split(":") for v in sys.argv[1:]] or [["513"], ["1025"], ["2049"], ["3073"]]
dev = torch.device("cuda:0")
ch, n = 4096, 1 << 20
x = torch.empty(ch, n, dtype=torch.float32, device=dev)
y = torch.empty_like(x)
filters.synth_f32(x, 1)
L = capi.lib()
def timed(fn, steps=5):
    fn(); fn(); torch.cuda.synchronize()
    t = L.llz_hip_timer_new(); L.llz_hip_timer_start(t, None)
    for _ in range(steps): fn()
    L.llz_hip_timer_stop(t, None); ms = L.llz_hip_timer_ms(t) / steps; L.llz_hip_timer_free(t)
    return ms
for case in cases:
    T = int(case[0]); algo = int(case[1]) if len(case) > 1 else 0
    taps = filters.fir_design("lpf", T, 0.1, 0.0, filters.KAISER)
    f = filters.FirFilterMC(ch, n, taps, algo=algo)
    ms = timed(lambda: f.filter(x, y))
    print(f"fir {T} taps algo {f.algo}: {ms:.2f} ms  {8 * ch * n / ms / 1e6:.0f} GB/s ({8 * ch * n / ms / 1e6 / 80:.1f} %)", flush=True)
    f.close()
