"""time the int16 1:3 decimator forms on BASELINE config 5's per-GPU shape (or smaller): python tools/time_i16.py [channels]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from llzlab_amd import capi, filters
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n = 3 * (((1 << 22) // 3) // 256 * 256)
dev = torch.device("cuda:0")
x = torch.empty(ch, n, dtype=torch.int16, device=dev)
y = torch.empty(ch, n // 3, dtype=torch.int16, device=dev)
filters.synth_i16(x, 0x11C0FFEE)
L = capi.lib()
def timed(fn, steps):
    fn(); torch.cuda.synchronize()
    t = L.llz_hip_timer_new(); L.llz_hip_timer_start(t, None)
    for _ in range(steps): fn()
    L.llz_hip_timer_stop(t, None); ms = L.llz_hip_timer_ms(t) / steps; L.llz_hip_timer_free(t)
    return ms
res = {}
for name, fmt, tune, steps in (("screened (bit-exact)", filters.PCM_I16, {}, 3), ("fast (1 LSB)", filters.PCM_I16_FAST, {}, 3),
                               ("all-double (bit-exact)", filters.PCM_I16, {"rs_i16_path": 1}, 1)):
    with capi.tuned(**tune):
        r = filters.ResampleMC(ch, 1, 3, 1.0, filters.BLACKMAN, fmt)
        ms = timed(lambda: r.process(x, y), steps)
        res[name] = y.clone() if "exact" in name else None
        print(f"{name:24s} {ch} ch x {n}: {ms:.2f} ms  {(2 + 2 / 3) * ch * n / ms / 1e6:.0f} GB/s ({(2 + 2 / 3) * ch * n / ms / 1e6 / 80:.1f} % of 8 TB/s)", flush=True)
        r.close()
a, b = res["screened (bit-exact)"], res["all-double (bit-exact)"]
print("screened == all-double on the whole batch:", bool(torch.equal(a, b)))
