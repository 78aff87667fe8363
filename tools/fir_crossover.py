import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from llzlab_amd import capi, filters
dev = torch.device("cuda:0"); L = capi.lib(); capi.check(L.llz_hip_set_device(0), "dev")
s = torch.cuda.current_stream(); sp = s.cuda_stream
for ch in (4096,):
    n = 1 << 20
    x = torch.empty(ch, n, dtype=torch.float32, device=dev); y = torch.empty_like(x)
    filters.synth_f32(x, 1, stream=s)
    for taps_n in (257, 513, 1025, 1537, 2049, 3073):
        taps = filters.fir_design("lpf", taps_n, 0.25, 0.0, filters.HAMMING)
        for algo in (1, 2, 3, 4, 5):
            if (algo == 1 and taps_n > 1025) or (algo == 5 and taps_n < 1025):
                continue
            try:
                f = filters.FirFilterMC(ch, n, taps, algo=algo, stream=s)
            except capi.LlzError:
                continue                                   # this many taps do not fit the algorithm
            f.filter(x, y); torch.cuda.synchronize()
            t = L.llz_hip_timer_new(); L.llz_hip_timer_start(t, sp)
            for _ in range(5): f.filter(x, y)
            L.llz_hip_timer_stop(t, sp); ms = L.llz_hip_timer_ms(t) / 5
            print(f"{ch}ch taps={taps_n} algo={algo}: {ms:.3f} ms {8*ch*n/ms/1e6:.0f} GB/s")
            f.close()
