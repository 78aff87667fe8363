#!/usr/bin/env python3
"""bench.py -- headline benchmark: multi-channel float32 FIR, 257 taps, 4096 channels per GPU (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by the driver under torch.distributed.run, one rank per GPU, backend nccl = RCCL)

A step = one pass of the hot path over one batch: llz_fir_filter_mc() on 4096 channels x 2^20 samples of
synthetic PCM that is already resident in HBM (generated on the device).  Channels shard across ranks with no
data-path collective (weak scaling: every GPU filters its own 4096 channels); the only exchange is the tap-table
broadcast from rank 0 at setup.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
BYTES_PER_SAMPLE = 8           # FIR fp32: 4 B read + 4 B written per sample-channel
CHANNELS = 4096
N_SAMPLES = 1 << 20
FLT_LEN = 257
SEED = 0x11C0FFEE


def host_cores():
    """(online cores by sysconf, cores this process may use): threads for the all-cores leg = the smaller of the two, cut
    further by a cgroup CPU quota when the box sets one"""
    online = os.sysconf("SC_NPROCESSORS_ONLN")
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else online
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = min(usable, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return online, max(1, min(online, usable))


def cpu_baseline(taps64, oracle_mod):
    """Reference CPU path (oracle/_ref when built, else the oracle port) on a bounded sample of the same workload: one
    llz_fir_filter state machine per channel, frame 4096 -- once on ONE core and once with the channels partitioned
    over all cores the box gives this process."""
    online, threads = host_cores()
    ch_per_thread, n = 24, 1 << 20                     # ~3-4 s of work per thread at ~7 Msamples/s/core
    kind = "reference" if oracle_mod.have_ref() else "port"
    orc = oracle_mod.Oracle()
    backend = oracle_mod.Ref() if kind == "reference" else orc
    frame = 4096
    x = orc.synth_f32(ch_per_thread, n, SEED).astype(np.float64)     # every thread filters the same 24 rows

    def work(_t):
        for c in range(ch_per_thread):
            # design inside the reference (same call the real caller makes), then stream frames
            backend.fir_stream(0, frame, FLT_LEN, 0.1, 0.0, 2, x[c], flush=False)

    def leg(nthreads):
        ths = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dt = time.perf_counter() - t0
        return nthreads * ch_per_thread * n / dt / 1e6, dt

    one, dt1 = leg(1)
    allc, dta = leg(threads)
    return {"value": allc, "unit": "Msamples/s", "cores": threads, "kind": kind,
            # NOT a whole-box figure when the process is given fewer cores than the host has: it is the rate of this slice
            "scope": (f"{threads}-thread slice of a {online}-core host (cores this process may use)" if threads < online
                      else f"all {online} cores of the host"),
            "sample": f"{threads} threads x {ch_per_thread} ch x {n} samples, {FLT_LEN}-tap llz_fir_filter, frame {frame} "
                      f"({dta:.1f} s); sysconf(_SC_NPROCESSORS_ONLN) = {online}",
            "single_core": {"value": one, "unit": "Msamples/s/core", "cores": 1,
                            "sample": f"1 thread x {ch_per_thread} ch x {n} samples ({dt1:.1f} s)"}}


def ols_kernel_name(channels, n):
    """the 1024-point overlap-save kernel (llzs_fir_ols_f32 in csrc/kernels/fir_ols.hip)"""
    return "k_fir_ols_chain_f32"


def kernel_source_sha():
    """sha256 over the headline kernel's sources: profiles/pmc_traffic.json records the one it was measured on, and
    `roofline.traffic` is reported only while they still match"""
    import hashlib
    h = hashlib.sha256()
    for f in ("fir_ols.hip", "fft32.hpp"):
        h.update(open(os.path.join(ROOT, "llzlab_amd", "csrc", "kernels", f), "rb").read())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--channels", type=int, default=CHANNELS, help="channels per GPU (weak scaling)")
    ap.add_argument("--samples", type=int, default=N_SAMPLES)
    ap.add_argument("--algo", type=int, default=0, help="0 auto (overlap-save), 1 time domain, 2 overlap-save, 3 time domain on the matrix cores, 4 / 5 overlap-save with\n"
                         "2048- / 4096-point transforms, 6 overlap-save with 8192-point transforms on pairs of waves")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--extra", action="store_true", help="(kept for old command lines: the extra configs now run by default)")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the channel-sharded resample / IIR configs (BASELINE configs 4 and 5) reported under 'also'")
    ap.add_argument("--dist-backend", default="nccl",
                    help="process-group backend; 'gloo' lets several ranks rehearse on ONE GPU (tables travel as CPU tensors)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from llzlab_amd import capi, filters, shard

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    dev_index = local_rank % torch.cuda.device_count()           # == local_rank on a real multi-GPU node
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    capi.check(capi.lib().llz_hip_set_device(dev_index), "llz_hip_set_device")
    comm_dev = dev if args.dist_backend == "nccl" else None       # where the tiny setup collectives live
    cpu_group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the coefficient tables travel over RCCL (backend nccl, device tensors); the timing barrier and the max-reduce of the
        # step time are host-side bookkeeping on a gloo group, which is also where a table goes should RCCL refuse to come up
        # (reported in the line as "table_broadcast")
        dist.init_process_group(backend=args.dist_backend, rank=rank, world_size=world)
        if args.dist_backend == "nccl":
            cpu_group = dist.new_group(backend="gloo")
            shard.use_cpu_group(cpu_group)

    def barrier():
        if world > 1:
            dist.barrier(group=cpu_group)

    channels, n = args.channels, args.samples
    # rank 0 designs the tap set (host C: llz_fir_lpf_cof(257, 0.1, KAISER)); everyone else receives it over RCCL
    taps = filters.fir_design("lpf", FLT_LEN, 0.1, 0.0, filters.KAISER) if rank == 0 else np.zeros(FLT_LEN)
    taps = shard.broadcast_table(taps, src=0, device=comm_dev)

    # every coefficient table the run needs is designed on rank 0 (host C code) and broadcast HERE, before anything is timed
    # and outside every guarded block: setup collectives either work on all ranks or end the job
    tables = {"fir257": taps}
    if not args.no_also:
        for t in (513, 1025, 2049, 3073):
            d = filters.fir_design("lpf", t, 0.1, 0.0, filters.KAISER) if rank == 0 else np.zeros(t)
            tables[f"fir{t}"] = shard.broadcast_table(d, src=0, device=comm_dev)
        mat = None
        if rank == 0:
            r0 = filters.ResampleMC(1, 1, 3, 1.0, filters.BLACKMAN, filters.PCM_F32)
            mat = r0.matrix()
            r0.close()
        shape = shard.broadcast_shape(mat.shape if rank == 0 else (0, 0), device=comm_dev)
        tables["rs_1to3"] = shard.broadcast_table(mat if rank == 0 else np.zeros(shape), device=comm_dev)
        for key, row in (("iir8_1024ch_sharded", [0.2066, 0.4131, 0.2066, 1.0, -0.3695, 0.1958]),
                         ("iir8_r099_1024ch_sharded", [0.01, 0.0, -0.01, 1.0, -2 * 0.99 * np.cos(0.3), 0.99 ** 2])):
            coef = np.tile(np.array(row), (8, 1)) if rank == 0 else np.zeros((8, 6))
            tables[key] = shard.broadcast_table(coef, device=comm_dev)

    stream = torch.cuda.current_stream()
    x = torch.empty(channels, n, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    filters.synth_f32(x, SEED, chan0=rank * channels, stream=stream)
    fir = filters.FirFilterMC(channels, n, taps, algo=args.algo, stream=stream)
    L = capi.lib()
    sptr = stream.cuda_stream

    for _ in range(args.warmup):
        fir.filter(x, y)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()

    # per-launch device time of the dominant kernel: HIP events on the stream the kernel runs on
    timers = [L.llz_hip_timer_new() for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        L.llz_hip_timer_start(timers[k], sptr)
        fir.filter(x, y)
        L.llz_hip_timer_stop(timers[k], sptr)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    wall = shard.max_over_ranks(wall, device=comm_dev)
    launch_ms = [L.llz_hip_timer_ms(t) for t in timers]
    for t in timers:
        L.llz_hip_timer_free(t)
    kern_ms = float(np.mean(launch_ms))

    fir_algo = fir.algo
    # the box's own streaming rate for the same 16 GiB -> 16 GiB: a device-to-device memcpy on the same stream
    memcpy_gbs = None
    if rank == 0:
        y.copy_(x)
        torch.cuda.synchronize()
        tm = L.llz_hip_timer_new()
        L.llz_hip_timer_start(tm, sptr)
        for _ in range(3):
            y.copy_(x)
        L.llz_hip_timer_stop(tm, sptr)
        memcpy_gbs = BYTES_PER_SAMPLE * channels * n / (L.llz_hip_timer_ms(tm) / 3 * 1e-3) / 1e9
        L.llz_hip_timer_free(tm)
        fir.filter(x, y)                                                   # restore y for the parity check below
        torch.cuda.synchronize()
    ms_per_step = wall / args.steps * 1e3
    samples_per_step = channels * n * world
    value = samples_per_step / (wall / args.steps) / 1e6                  # Msamples/s, whole job

    # parity against the oracle (outside the timed region), SURVEY.md 8(d): first 4 + last 4 channels over their FULL
    # length (every segment hand-over of the chain kernel, the ragged last segment) and all channels x first 16 Ki samples
    parity = None
    also = {}
    cpu = None
    orc = None
    if rank == 0:
        from oracle import pyoracle
        orc = pyoracle.Oracle()
        keep = FLT_LEN - 1
        h64 = taps.astype(np.float32).astype(np.float64)

        def rms_pair(got, ref):
            err = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)))
            return err, err / float(np.sqrt(np.mean(ref ** 2)))

        # the handle streams: every step after the first starts from the previous step's last 256 samples, so the
        # oracle is fed [tail of x | x] and its first 256 outputs are dropped
        sel = sorted(set(list(range(min(4, channels))) + list(range(max(0, channels - 4), channels))))
        xs = torch.cat([x[sel, n - keep:], x[sel]], dim=1).cpu().numpy()
        full_abs, full_rel = rms_pair(y[sel].cpu().numpy(), orc.fir_batch_f32_mt(xs, h64)[:, keep:])
        m = min(n, 1 << 14)
        xs = torch.cat([x[:, n - keep:], x[:, :m]], dim=1).cpu().numpy()
        head_abs, head_rel = rms_pair(y[:, :m].cpu().numpy(), orc.fir_batch_f32_mt(xs, h64)[:, keep:])
        del xs
        parity = {"rms_abs": max(full_abs, head_abs), "rms_rel": max(full_rel, head_rel), "tolerance": 1e-5,
                  "full_length": {"channels": sel, "samples": n, "rms_abs": full_abs, "rms_rel": full_rel},
                  "all_channels_head": {"channels": channels, "samples": m, "rms_abs": head_abs, "rms_rel": head_rel},
                  "checked": f"{len(sel)} ch x {n} samples + {channels} ch x {m} samples vs CPU oracle "
                             "(streaming state included)"}
        if world == 1 and not args.no_cpu:
            cpu = cpu_baseline(taps, pyoracle)

    # ---- 'also' configurations (never part of `value`) ----
    # A block times its configurations LOCALLY (no collective inside, so a rank that fails cannot leave its peers waiting in
    # one); after the block all ranks agree in ONE all-reduce (MAX) on an error flag and the per-configuration times.  A block
    # that failed on any rank is reported as an error by every rank and the run goes on to the headline line.
    agree_group = cpu_group                                   # None = the default group (gloo rehearsal runs)

    def run_block(label, keys, fn):
        entries, err = {}, None
        try:
            entries = fn()                                    # {key: (local ms, make(ms) -> dict)}
            missing = [k for k in keys if k not in entries]
            if missing:
                raise RuntimeError("block returned no timing for " + ", ".join(missing))
        except Exception as e:  # noqa: BLE001
            err = str(e).splitlines()[0][:200] if str(e) else type(e).__name__
            sys.stderr.write(f"bench.py: also/{label} failed on rank {rank}: {err}\n")
        ms = [entries[k][0] if err is None else 0.0 for k in keys]
        if world > 1:
            vec = torch.tensor([1.0 if err else 0.0] + ms, dtype=torch.float64)
            dist.all_reduce(vec, op=dist.ReduceOp.MAX, group=agree_group)
            if vec[0].item() > 0:
                return {label: {"error": err or "failed on another rank"}}
            ms = vec[1:].tolist()
        elif err:
            return {label: {"error": err}}
        return {k: entries[k][1](m) for k, m in zip(keys, ms)}

    def time_local(fn, steps, warm=1):
        """average device time of `steps` calls on this rank's stream (HIP events), after `warm` untimed calls"""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        tm = L.llz_hip_timer_new()
        L.llz_hip_timer_start(tm, sptr)
        for _ in range(steps):
            fn()
        L.llz_hip_timer_stop(tm, sptr)
        ms = L.llz_hip_timer_ms(tm) / steps
        L.llz_hip_timer_free(tm)
        return ms

    ctx = dict(torch=torch, filters=filters, capi=capi, shard=shard, dev=dev, stream=stream, rank=rank, world=world,
               time_local=time_local, orc=(orc if rank == 0 else None), tables=tables)

    # long filters on the same batch (weak scaling like the headline): 513 taps on the 2048-point overlap-save, 1025 on the
    # 4096-point one, 2049 and 3073 on the 8192-point one (the library's own choice)
    LONG = (513, 1025, 2049, 3073)

    def long_fir():
        res = {}
        for long_taps in LONG:
            lf = filters.FirFilterMC(channels, n, tables[f"fir{long_taps}"], stream=stream)
            lms = time_local(lambda: lf.filter(x, y), 4, warm=2)
            algo = {4: "overlap-save-2048", 5: "overlap-save-4096", 6: "overlap-save-8192"}.get(lf.algo, str(lf.algo))
            lf.close()
            res[f"fir_{long_taps}taps_{channels}ch_per_gpu"] = (lms, lambda ms, algo=algo: {
                "Msamples_s": channels * n * world / ms / 1e3, "GBs_per_gpu": BYTES_PER_SAMPLE * channels * n / ms / 1e6,
                "hbm_frac_per_gpu": BYTES_PER_SAMPLE * channels * n / ms / 1e6 / HBM_PEAK_GBS, "ms": ms,
                "channels_per_gpu": channels, "scaling": "weak", "algorithm": algo})
        return res

    if not args.no_also and fir_algo == 2:
        also.update(run_block("long_fir", [f"fir_{t}taps_{channels}ch_per_gpu" for t in LONG], long_fir))

    # release the FIR batch before the other configs allocate theirs
    fir.close()
    del x, y
    torch.cuda.empty_cache()
    if not args.no_also:
        for label, keys, fn in sharded_blocks(ctx):
            also.update(run_block(label, keys, fn))
            torch.cuda.empty_cache()
    if not args.no_also and world == 1:
        also.update(run_block("extra_paths", EXTRA_KEYS, lambda: extra_paths(ctx)))
        torch.cuda.empty_cache()
        if torch.cuda.device_count() > 1:
            also.update(run_block("sharded_c_abi", ["sharded_c_abi"], lambda: sharded_c_abi(ctx)))

    if rank == 0:
        achieved = BYTES_PER_SAMPLE * channels * n / (kern_ms * 1e-3) / 1e9      # GB/s, algorithmic bytes
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        # the committed PMC figure is per launch of the DEFAULT workload: report it only for that workload
        if os.path.exists(tpath) and channels == CHANNELS and n == N_SAMPLES and fir_algo == 2:
            try:
                rec = json.load(open(tpath))
                # measured on exactly these kernel sources?  (a stale figure is not reported)
                if rec.get("kernel_source_sha256") == kernel_source_sha():
                    traffic = rec.get("headline_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        algo_name = {1: "time-domain", 2: "overlap-save-1024", 3: "time-domain-matrix-core", 4: "overlap-save-2048",
                     5: "overlap-save-4096"}[fir_algo]
        line = {
            "metric": "Msamples/s/GPU (float32 FIR 257-tap, 4096 ch) + achieved HBM GB/s vs peak",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "table_broadcast": shard.transport() if world > 1 else "single process",
            "config": {"workload": f"{channels}-ch float32 FIR, {FLT_LEN} taps (llz_fir_lpf_cof 0.1 KAISER), "
                                   f"{n} samples/ch per GPU, {algo_name}, channels sharded over {world} GPU(s)",
                       "channels_per_gpu": channels, "samples_per_channel": n, "taps": FLT_LEN,
                       "algorithm": algo_name, "per_gpu_Msamples_s": value / world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": {1: "k_fir_td_f32", 2: ols_kernel_name(channels, n), 3: "k_fir_mfma_bf16x3",
                                    4: "k_fir_ols2k_chain_f32", 5: "k_fir_ols4k_f32", 6: "k_fir_ols8k_f32"}[fir_algo],
                         "kernel_ms_avg": kern_ms, "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE * channels * n,
                         "memcpy_d2d_GBs": memcpy_gbs,
                         "frac_of_memcpy_d2d": (achieved / memcpy_gbs) if memcpy_gbs else None},
            "cpu_baseline": cpu,
            "parity": parity,
            # north_star's scaling target (>= 6x at 8 GPUs) is on the channel-sharded RESAMPLE, a fixed 8192 channels over the
            # ranks (strong scaling): that curve is this entry's Msamples_in_s per N, not `value` (weak-scaled FIR)
            "strong_scaling_metric": "also.resample_1to3_f32_8192ch_sharded.Msamples_in_s",
        }
        # measured HBM traffic over algorithmic bytes of the kernels behind the 'also' entries (profiles/pmc_traffic.json:
        # FETCH_SIZE x 2 + WRITE_SIZE per launch of tools/run_path.py traffic_set), reported while the kernel sources still match
        try:
            recs = json.load(open(tpath)).get("kernels", {}) if os.path.exists(tpath) else {}
        except Exception:
            recs = {}
        for key, kern in (("resample_1to3_f32_8192ch_sharded", "k_fir_mfma_bf16x3#0"), ("resample_1to3_i16_exact_8192ch_sharded", "k_fir_mfma_i8x#0"),
                          ("resample_1to3_i16_fast_8192ch_sharded", "k_fir_mfma_i8x#0"), ("iir8_1024ch_sharded", "k_iir_cascade_wave_pk32#0"),
                          ("iir8_r099_1024ch_sharded", "k_iir_cascade_wave_pf64w#0"), ("resample_147to160_f32_256ch", "k_resample_mfma_pt_f32#0"),
                          ("resample_160to147_f32_256ch", "k_resample_mfma_pt_f32#1"), ("resample_147to160_i16_256ch", "k_resample_i8d#0"),
                          ("resample_160to147_i16_256ch", "k_resample_i8d#1"), ("mdct_fixed_fwd_2048x65536", "k_mdct4_q15#0"),
                          ("mdct_fixed_inv_2048x65536", "k_mdct4_q15#1"), ("mdct_frames_analysis_256x1024ch", "k_mdct_reg_f32#0"),
                          ("mdct_frames_synthesis_256x1024ch", "k_mdct_reg_f32#1")):
            rec = recs.get(kern)
            if rec and key in also and "error" not in also[key]:
                files = {"k_iir": ["iir.hip"], "k_fir_mfma_bf16x3": ["fir_mfma.hip"], "k_fir_mfma_i16": ["fir_mfma.hip"],
                         "k_fir_mfma_i8x": ["fir_mfma_i8.hip", "screen_i8.hpp"], "k_resample_mfma": ["resample_mfma.hip"],
                         "k_resample_i8d": ["resample_i8.hip", "screen_i8.hpp"], "k_mdct4_q15": ["mdct_q15.hip", "fft_core.hpp"],
                         "k_mdct_reg_f32": ["fft.hip", "fft_core.hpp"]}
                src = next(v for p_, v in files.items() if kern.startswith(p_))
                import hashlib
                h = hashlib.sha256()
                for fn in src:
                    h.update(open(os.path.join(ROOT, "llzlab_amd", "csrc", "kernels", fn), "rb").read())
                also[key]["kernel"] = kern.split("#")[0]
                also[key]["traffic_over_algorithmic"] = (rec["traffic_over_algorithmic"]
                                                         if rec.get("kernel_source_sha256") == h.hexdigest() else None)
        if also:
            line["also"] = also
        print(json.dumps(line), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


def _spread(channels, k=8):
    """k channel indices spread over [0, channels)"""
    return sorted(set(int(round(i * (channels - 1) / max(k - 1, 1))) for i in range(min(k, channels))))


def sharded_blocks(ctx):
    """BASELINE configs 5 and 4 as the north star states them: a FIXED total channel count sharded over the ranks (strong
    scaling); the coefficient tables were designed on rank 0 and broadcast at setup (ctx["tables"]).  Reported under 'also';
    the headline 'value' is the FIR.  Returns (label, keys, fn) blocks for run_block: fn times locally and returns
    {key: (ms, make)}."""
    torch, filters, shard = ctx["torch"], ctx["filters"], ctx["shard"]
    dev, stream, rank, world, time_local, orc, tables = (ctx[k] for k in ("dev", "stream", "rank", "world", "time_local",
                                                                          "orc", "tables"))

    # config 5: 8192-ch polyphase resample 48 kHz -> 16 kHz (L=1, M=3), 4 Mi samples/ch
    RS_CH = 8192
    rs_n = 3 * (((1 << 22) // 3) // 256 * 256)            # 4 Mi rounded down to whole 3:1 periods x 256

    def rs_entry(total_ch, ch, n, bytes_per_in, extra=None):
        def make(ms):
            e = {"Msamples_in_s": total_ch * n / ms / 1e3, "GBs_per_gpu": bytes_per_in * ch * n / ms / 1e6,
                 "hbm_frac_per_gpu": bytes_per_in * ch * n / ms / 1e6 / HBM_PEAK_GBS, "ms": ms, "channels_per_gpu": ch,
                 "scaling": "strong"}
            e.update(extra or {})
            return e
        return make

    def resample_f32():
        lo, hi = shard.channel_range(RS_CH, rank, world)
        ch = hi - lo
        x = torch.empty(ch, rs_n, dtype=torch.float32, device=dev)
        y = torch.empty(ch, rs_n // 3, dtype=torch.float32, device=dev)
        filters.synth_f32(x, SEED, chan0=lo, stream=stream)
        r = filters.ResampleMC(ch, 1, 3, 1.0, filters.BLACKMAN, filters.PCM_F32, stream=stream)
        r.set_matrix(tables["rs_1to3"])                     # every rank runs with rank 0's tap matrix
        ms = time_local(lambda: r.process(x, y), 3)
        r.close()
        return {"resample_1to3_f32_8192ch_sharded": (ms, rs_entry(RS_CH, ch, rs_n, 4 + 4 / 3))}

    # config 5, int16 PCM in and out (the reference resampler's own sample format): the bit-exact form (screened on the matrix
    # cores, the reference's double order where the screen cannot decide) and the matrix-core fp32-accumulate form (within 1
    # LSB), same sharding and matrix.  `parity` is COMPUTED here (rank 0, outside the timed region): spread channels x the
    # first 64 Ki outputs against the oracle's llz_resample loop.
    def resample_i16():
        lo, hi = shard.channel_range(RS_CH, rank, world)
        ch = hi - lo
        xi = torch.empty(ch, rs_n, dtype=torch.int16, device=dev)
        yi = torch.empty(ch, rs_n // 3, dtype=torch.int16, device=dev)
        filters.synth_i16(xi, SEED, chan0=lo, stream=stream)
        out = {}
        for key, fmt, steps in (("resample_1to3_i16_exact_8192ch_sharded", filters.PCM_I16, 2),
                                ("resample_1to3_i16_fast_8192ch_sharded", filters.PCM_I16_FAST, 3)):
            def handle(fmt=fmt):
                r = filters.ResampleMC(ch, 1, 3, 1.0, filters.BLACKMAN, fmt, stream=stream)
                r.set_matrix(tables["rs_1to3"])
                return r
            r = handle()
            ms = time_local(lambda: r.process(xi, yi), steps)
            r.close()
            out[key] = (ms, rs_entry(RS_CH, ch, rs_n, 2 + 2 / 3, {"parity": i16_parity(orc, handle, xi, yi, 1, 3, 1 << 16)}))
        return out

    # config 4: 1024-ch IIR, 8-biquad cascade, 1 Mi samples/ch; two coefficient sets: 8 copies of the
    # in-tree low-pass section (pole radius 0.44, float32 arithmetic passes the noise-gain check) and 8 high-Q sections at
    # pole radius 0.99 (double arithmetic)
    IIR_CH = 1024

    def iir():
        lo, hi = shard.channel_range(IIR_CH, rank, world)
        ch = hi - lo
        n = 1 << 20
        x = torch.empty(ch, n, dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        filters.synth_f32(x, SEED, chan0=lo, stream=stream)
        out = {}
        for key in ("iir8_1024ch_sharded", "iir8_r099_1024ch_sharded"):
            q = filters.IirCascadeMC(ch, tables[key], stream=stream)
            ms = time_local(lambda: q.filter(x, y), 20, warm=10)   # a 2 ms kernel: clocks need tens of ms to settle after idling
            prec = q.precision
            q.close()

            def make(ms, prec=prec, ch=ch):
                # FMA equivalents per sample and section: 6 in the float32 kernel (b0 folded into one input gain), 7 in double
                fma = 6 if prec == 32 else 7
                peak = 157.3 if prec == 32 else 78.6
                return {"Msamples_s": IIR_CH * n / ms / 1e3, "GBs_per_gpu": 8 * ch * n / ms / 1e6,
                        "hbm_frac_per_gpu": 8 * ch * n / ms / 1e6 / HBM_PEAK_GBS, "ms": ms, "channels_per_gpu": ch,
                        "scaling": "strong",
                        # the binding roof of this kernel is the vector pipe, not HBM
                        "arithmetic": "f32" if prec == 32 else "f64",
                        "valu_TFLOPs_per_gpu": 2 * fma * 8 * ch * n / ms / 1e9, "valu_peak_TFLOPs": peak,
                        "valu_frac_per_gpu": 2 * fma * 8 * ch * n / ms / 1e9 / peak}
            out[key] = (ms, make)
        return out

    return [("resample_f32", ["resample_1to3_f32_8192ch_sharded"], resample_f32),
            ("resample_i16", ["resample_1to3_i16_exact_8192ch_sharded", "resample_1to3_i16_fast_8192ch_sharded"], resample_i16),
            ("iir", ["iir8_1024ch_sharded", "iir8_r099_1024ch_sharded"], iir)]


def i16_parity(orc, fresh_handle, xi, yi, L_, M_, n_out_check, gain=1.0, win=1):
    """compare spread channels x the first n_out_check outputs of an int16 resampler with the oracle's llz_resample loop (rank
    0 only; None elsewhere).  The timed handle has streamed several calls (its history is no longer zero), so ONE more call is
    made here on a fresh handle (fresh_handle() -> ResampleMC), outside every timed region."""
    if orc is None:
        return None
    import torch
    r = fresh_handle()
    r.process(xi, yi)
    torch.cuda.synchronize()
    r.close()
    info = orc.rs_info(2, L_, M_, gain, win)
    nin, nout = info["bytes_in"] // 2, info["bytes_out"] // 2
    frames = max(1, min(n_out_check // nout, xi.shape[1] // nin))
    sel = _spread(xi.shape[0])
    ref = orc.rs_batch_i16(xi[sel, : frames * nin].cpu().numpy(), L_, M_, gain, win)
    got = yi[sel, : frames * nout].cpu().numpy()
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    return {"checked": f"{len(sel)} ch x {frames * nout} outputs vs oracle llz_resample", "mismatches": int((diff != 0).sum()),
            "max_abs_diff_lsb": int(diff.max()), "bit_exact": bool((diff == 0).all())}


EXTRA_KEYS = ["fir63_64ch_time_domain", "fir63_64ch_overlap_save", "resample_147to160_f32_256ch", "resample_160to147_f32_256ch",
              "resample_147to160_i16_256ch", "resample_160to147_i16_256ch", "mdct_fixed_fwd_2048x65536", "mdct_fixed_inv_2048x65536",
              "mdct_frames_analysis_256x1024ch", "mdct_frames_synthesis_256x1024ch", "autocorr_direct_p16_262144x1024",
              "iir_order3_df1_1024ch"]


def extra_paths(ctx):
    """Other BASELINE.json configs on one GPU, few steps each (context, not the headline)."""
    torch, filters, dev, stream, time_local, orc = (ctx[k] for k in ("torch", "filters", "dev", "stream", "time_local", "orc"))
    out = {}

    # config 2: 64 ch x 63 taps x 1 Mi on its stated algorithm (time domain) and on the library's own choice
    ch, n = 64, 1 << 20
    x = torch.empty(ch, n, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    filters.synth_f32(x, SEED, stream=stream)
    taps63 = filters.fir_design("lpf", 63, 0.25, 0.0, filters.HAMMING)
    for name, algo in (("fir63_64ch_time_domain", filters.FIR_ALGO_TIME), ("fir63_64ch_overlap_save", filters.FIR_ALGO_OVERLAP_SAVE)):
        f = filters.FirFilterMC(ch, n, taps63, algo=algo, stream=stream)
        ms = time_local(lambda: f.filter(x, y), 10)
        out[name] = (ms, lambda ms, ch=ch, n=n: {"Msamples_s": ch * n / ms / 1e3, "GBs": 8 * ch * n / ms / 1e6,
                                                  "hbm_frac": 8 * ch * n / ms / 1e6 / HBM_PEAK_GBS, "ms": ms})
        f.close()
    del x, y
    # the reference CLI's default ratio 147:160 (48 kHz -> 44.1 kHz) and its inverse, 256 channels (general L/M path): float32,
    # and the reference's own int16 format, bit-exact (parity computed against the oracle, outside the timed region)
    for (L_, M_) in ((147, 160), (160, 147)):
        ch, n = 256, M_ * 8192
        x = torch.empty(ch, n, dtype=torch.float32, device=dev)
        y = torch.empty(ch, n * L_ // M_, dtype=torch.float32, device=dev)
        filters.synth_f32(x, SEED, stream=stream)
        r = filters.ResampleMC(ch, L_, M_, 1.0, filters.BLACKMAN, filters.PCM_F32, stream=stream)
        ms = time_local(lambda: r.process(x, y), 20)
        r.close()

        def make(ms, ch=ch, n=n, bpi=4 + 4 * L_ / M_, extra=None):
            gb = bpi * ch * n / ms / 1e6
            e = {"Msamples_in_s": ch * n / ms / 1e3, "GBs": gb, "hbm_frac": gb / HBM_PEAK_GBS, "ms": ms}
            e.update(extra or {})
            return e
        out[f"resample_{L_}to{M_}_f32_256ch"] = (ms, make)
        del x, y
        xi = torch.empty(ch, n, dtype=torch.int16, device=dev)
        yi = torch.empty(ch, n * L_ // M_, dtype=torch.int16, device=dev)
        filters.synth_i16(xi, SEED, stream=stream)
        def handle(L_=L_, M_=M_, ch=ch):
            return filters.ResampleMC(ch, L_, M_, 1.0, filters.BLACKMAN, filters.PCM_I16, stream=stream)
        r = handle()
        ms = time_local(lambda: r.process(xi, yi), 5)
        r.close()
        par = i16_parity(orc, handle, xi, yi, L_, M_, 1 << 16)
        out[f"resample_{L_}to{M_}_i16_256ch"] = (ms, lambda ms, ch=ch, n=n, bpi=2 + 2 * L_ / M_, par=par, make=make:
                                                  make(ms, ch, n, bpi, {"parity": par}))
        del xi, yi
    # the fixed-point MDCT in batch (int32, Q15 tables, bit-exact: one launch per direction) and the windowed
    # 50 %-overlap MDCT frames (float32; the synthesis writes every sample once)
    n, count = 2048, 1 << 16
    xq = torch.randint(-(1 << 20), 1 << 20, (count, n), dtype=torch.int32, device=dev)
    Xq = torch.empty(count, n // 2, dtype=torch.int32, device=dev)
    mq = filters.MdctFixed(2, n)
    mq.set_stream(stream)
    for name, fn in (("mdct_fixed_fwd_2048x65536", lambda: mq.forward_batch(xq, Xq)),
                     ("mdct_fixed_inv_2048x65536", lambda: mq.inverse_batch(Xq, xq))):
        ms = time_local(fn, 10)
        out[name] = (ms, lambda ms, b=6 * n * count: {"GBs": b / ms / 1e6, "hbm_frac": b / ms / 1e6 / HBM_PEAK_GBS, "ms": ms,
                                                        "bytes_per_sample": 6})
    mq.close()
    del xq, Xq
    F, ch, frames = 256, 1024, 1024
    xf = torch.rand(ch, frames * F, dtype=torch.float32, device=dev) * 2 - 1
    Xf = torch.empty(ch, frames, F, dtype=torch.float32, device=dev)
    mf = filters.MdctFramesMC(ch, F, 0, stream=stream)
    for name, fn in (("mdct_frames_analysis_256x1024ch", lambda: mf.analysis(xf, Xf)),
                     ("mdct_frames_synthesis_256x1024ch", lambda: mf.synthesis(Xf, xf))):
        ms = time_local(fn, 10)
        out[name] = (ms, lambda ms, b=8 * F * ch * frames: {"GBs": b / ms / 1e6, "hbm_frac": b / ms / 1e6 / HBM_PEAK_GBS, "ms": ms,
                                                             "bytes_per_sample": 8})
    mf.close()
    del xf, Xf
    # direct autocorrelation at an LPC order (float32 batch form of llz_autocorr), and the reference's own
    # order-3 direct-form-I filter (libllzaudio/llz_musicpitch.c:1277-1285) on 1024 channels (llz_iir_mc: exact-order double per lane)
    frames_ac, n_ac, p_ac = 1 << 18, 1024, 16
    xa = torch.rand(frames_ac, n_ac, dtype=torch.float32, device=dev) * 2 - 1
    ra = torch.empty(frames_ac, p_ac + 1, dtype=torch.float32, device=dev)
    ms = time_local(lambda: filters.autocorr_mc(xa, ra, p_ac, stream=stream), 10)
    out["autocorr_direct_p16_262144x1024"] = (ms, lambda ms, b=4 * frames_ac * n_ac: {"GBs": b / ms / 1e6, "hbm_frac": b / ms / 1e6 / HBM_PEAK_GBS,
                                                                                      "ms": ms, "bytes_per_sample": 4})
    del xa, ra
    ch, n = 1024, 1 << 20
    xi3 = torch.empty(ch, n, dtype=torch.float32, device=dev)
    yi3 = torch.empty_like(xi3)
    filters.synth_f32(xi3, SEED, stream=stream)
    q3 = filters.IirMC(ch, [1.0, -0.3695, 0.1958, 0.0], [1.0, 0.2066, 0.4131, 0.2066], stream=stream)
    ms = time_local(lambda: q3.filter(xi3, yi3), 5)
    out["iir_order3_df1_1024ch"] = (ms, lambda ms, b=8 * ch * n: {"Msamples_s": ch * n / ms / 1e3, "GBs": b / ms / 1e6,
                                                                    "hbm_frac": b / ms / 1e6 / HBM_PEAK_GBS, "ms": ms})
    q3.close()
    del xi3, yi3
    return out


def sharded_c_abi(ctx):
    """The library's OWN multi-GPU path (include/llz_shard.h: one process, one handle, ncclCommInitAll + ncclBroadcast of the
    tables at init), reachable when one process sees several GPUs: BASELINE config 5 (8192 ch, 1:3, float32) through
    llz_resample_mc_sharded over all visible devices, timed with llz_sharded_timer_* (per-GPU events, max reported)."""
    torch, filters, capi = ctx["torch"], ctx["filters"], ctx["capi"]
    ndev = torch.cuda.device_count()
    devices = list(range(ndev))
    total_ch = 8192
    n = 3 * (((1 << 22) // 3) // 256 * 256)
    h = filters.ResampleMCSharded(total_ch, 1, 3, 1.0, filters.BLACKMAN, filters.PCM_F32, devices)
    xs, ys = h.alloc(n, torch.float32), h.alloc(n // 3, torch.float32)
    for (d, c0, _cnt), x in zip(h.shards, xs):
        with torch.cuda.device(d):
            capi.check(capi.lib().llz_hip_set_device(d), "llz_hip_set_device")
            filters.synth_f32(x, SEED, chan0=c0, stream=torch.cuda.current_stream(d))
            torch.cuda.synchronize(d)
    capi.check(capi.lib().llz_hip_set_device(ctx["dev"].index), "llz_hip_set_device")
    h.process(xs, ys)
    h.synchronize()
    steps = 3
    h.timer_start()
    for _ in range(steps):
        h.process(xs, ys)
    h.timer_stop()
    ms, per = h.timer_ms()
    ms, per = ms / steps, [float(v) / steps for v in per]
    ranks = h.rccl_ranks
    h.close()

    def make(ms):
        return {"workload": f"{total_ch}-ch resample 1:3 float32, {n} samples/ch, one process, {ndev} GPUs through "
                            "llz_resample_mc_sharded", "Msamples_in_s": total_ch * n / ms / 1e3, "ms": ms,
                "per_shard_ms": per, "n_devices": ndev, "rccl_ranks": ranks, "scaling": "strong"}
    return {"sharded_c_abi": (ms, make)}


if __name__ == "__main__":
    main()
