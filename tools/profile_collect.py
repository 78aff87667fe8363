#!/usr/bin/env python3
"""Turn gpurun_out/profile_<tag>/ into the committed summaries under profiles/ (kernel stats csv, bench line,
PMC traffic json, SQ counter table)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "profile_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: take the latest run's file, not the first match"""
    return max(glob.glob(pattern), key=os.path.getmtime)


shutil.copy(newest(src + "/stats/*/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
for name in ("bench.json", "bench_under_rocprof.json"):
    line = [l for l in open(os.path.join(src, name)) if l.startswith("{")][0]
    open(os.path.join(dst, f"{tag}_{name}"), "w").write(line)
bench = json.loads([l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][0])
kern = bench["roofline"]["kernel"]

# warm launches only: the kernel trace of the profiled run lists every dispatch; the timed region of bench.py is dispatches
# warmup .. warmup + steps - 1 of the headline kernel (the first `warmup` are untimed, later ones belong to the memcpy /
# parity legs)
prof = json.loads([l for l in open(os.path.join(src, "bench_under_rocprof.json")) if l.startswith("{")][0])
trace = newest(src + "/stats/*/*_kernel_trace.csv")
durs = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        for r in csv.DictReader(open(trace)) if kern in r["Kernel_Name"]]
durs = [d for _t, d in sorted(durs)]
w, k = prof["warmup"], prof["steps"]
timed = durs[w:w + k]
warm = {"kernel": kern, "command": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu  (steps {k}, warmup {w})",
        "dispatches": len(durs), "all_ms": [round(d, 4) for d in durs],
        "timed_region_avg_ms": sum(timed) / len(timed), "timed_region_min_ms": min(timed), "timed_region_max_ms": max(timed),
        "bench_line_kernel_ms_avg_same_run": prof["roofline"]["kernel_ms_avg"],
        "bench_line_frac_same_run": prof["roofline"]["frac"],
        "frac_from_trace": prof["roofline"]["algorithmic_bytes_per_launch"] / (sum(timed) / len(timed) * 1e-3) / 1e9 / 8000.0}
json.dump(warm, open(os.path.join(dst, f"{tag}_headline_kernel_trace.json"), "w"), indent=1)


def counters(sub):
    """{(kernel name, grid size, run): {counter: [value per dispatch]}} plus the dispatch durations (ms) under the key "_ms".
    One kernel name can be launched with several grids in one run (the headline batch and the 64-channel config share
    k_fir_ols_chain_f32): dispatches are told apart by (name, grid), never averaged across grids."""
    f = newest(f"{src}/{sub}/*/*_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    seen = set()
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"], r["Grid_Size"])
        agg[key][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        if (key, r["Dispatch_Id"]) not in seen:
            seen.add((key, r["Dispatch_Id"]))
            agg[key]["_ms"].append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
            agg[key]["_id"].append((int(r["Dispatch_Id"]), int(r["Dispatch_Id"])))
    # two workloads of one kernel may share name AND grid (147:160 and 160:147 on k_resample_i8d): a workload's launches follow
    # each other within a few dispatches, so a group is split where the dispatch ids jump
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for key, cols in agg.items():
        ids = sorted(i for i, _ in cols["_id"])
        run_of, run = {}, 0
        for a, b in zip([ids[0]] + ids, ids):
            run += b - a > 8
            run_of[b] = run
        for c, vals in cols.items():
            for i, v in vals:
                out[(key[0], key[1], run_of[i])][c].append(v)
    return out


def pick(agg, needle, grid=None):
    """per-launch averages of the dispatches of ONE (kernel, grid): the grid asked for, else the heaviest group (longest mean
    duration) -- for the headline kernel that is the 4096-channel batch"""
    groups = [(k, v) for k, v in agg.items() if needle in k[0] and (grid is None or str(grid) == k[1])]
    if not groups:
        return {}, {}
    k, v = max(groups, key=lambda kv: sum(kv[1]["_ms"]) / len(kv[1]["_ms"]))
    avg = {c: sum(x) / len(x) for c, x in v.items()}
    avg["_grid"] = int(k[1])
    return avg, {c: len(x) for c, x in v.items()}


def by_name(agg):
    """{kernel name: {counter: [values]}} of each name's heaviest grid (for the per-kernel utilisation table)"""
    out = {}
    for name in {k[0] for k in agg}:
        groups = [(k, v) for k, v in agg.items() if k[0] == name]
        out[name] = max(groups, key=lambda kv: sum(kv[1]["_ms"]) / len(kv[1]["_ms"]))[1]
    return out


fetch, nf = pick(counters("pmc_fetch"), kern)
write, nw = pick(counters("pmc_write"), kern)
tailf, _ = pick(counters("pmc_fetch"), "k_fir_tail_f32")
synw, _ = pick(counters("pmc_write"), "k_synth_f32")
out = {
    "kernel": kern,
    "grid": fetch.get("_grid"),
    "FETCH_SIZE_KB_per_launch_raw": fetch["FETCH_SIZE"], "FETCH_SIZE_launches": nf["FETCH_SIZE"],
    "WRITE_SIZE_KB_per_launch_raw": write["WRITE_SIZE"], "WRITE_SIZE_launches": nw["WRITE_SIZE"],
    "fetch_bytes_per_launch_corrected_x2": 2 * fetch["FETCH_SIZE"] * 1024,
    "write_bytes_per_launch": write["WRITE_SIZE"] * 1024,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "kernel_source_sha256": __import__("hashlib").sha256(
        b"".join(open(os.path.join(root, "llzlab_amd", "csrc", "kernels", f), "rb").read() for f in ("fir_ols.hip", "fft32.hpp"))
    ).hexdigest(),
    "calibration": {
        "k_fir_tail_f32_FETCH_SIZE_KB_raw": tailf.get("FETCH_SIZE"),
        "k_fir_tail_f32_true_read_bytes": 4096 * 256 * 4,
        "k_synth_f32_WRITE_SIZE_KB_raw": synw.get("WRITE_SIZE"),
        "k_synth_f32_true_write_bytes": 4096 * (1 << 20) * 4,
    },
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 3 "
            "--warmup 1 --no-cpu`; counter unit KB; FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes "
            "for gfx950 -- the factor is re-checked here on k_fir_tail_f32 (known read bytes, dword-per-lane coalesced "
            "like the headline kernel) and WRITE_SIZE on k_synth_f32 (known write bytes)",
}
out["headline_kernel_bytes_per_launch"] = out["fetch_bytes_per_launch_corrected_x2"] + out["write_bytes_per_launch"]
json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
# the bench lines of this round were printed before this round's PMC passes existed: give them this round's traffic figure
for name in ("bench.json", "bench_under_rocprof.json"):
    path = os.path.join(dst, f"{tag}_{name}")
    line = json.loads(open(path).read())
    line["roofline"]["traffic"] = out["headline_kernel_bytes_per_launch"]
    open(path, "w").write(json.dumps(line) + "\n")
sq, per_pass = {}, {}
for sub in ("pmc_sq1", "pmc_sq2"):
    v, cnt = pick(counters(sub), kern)
    per_pass[sub] = {"grid": v.pop("_grid", None), "launch_ms_in_this_pass": v.pop("_ms", None), "launches": max(cnt.values())}
    sq.update(v)
# what the counters say about the vector pipe (units: SQ_INSTS_* count wave-instructions; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES /
# SQ_WAIT_* count quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs -- MI355X_MICROARCH.md)
SIMDS = 1024
derived = {}
if "GRBM_GUI_ACTIVE" in sq:
    cyc = sq["GRBM_GUI_ACTIVE"] / 8
    ms2 = per_pass["pmc_sq2"]["launch_ms_in_this_pass"]
    derived["shader_cycles_per_launch"] = cyc
    derived["shader_clock_GHz_during_kernel"] = cyc / (ms2 * 1e6)
    if "SQ_INSTS_VALU" in sq:
        derived["valu_wave_instructions_per_simd"] = sq["SQ_INSTS_VALU"] / SIMDS
        derived["shader_cycles_per_valu_instruction_per_simd"] = cyc / (sq["SQ_INSTS_VALU"] / SIMDS)
    if "SQ_ACTIVE_INST_VALU" in sq:
        # measured in the OTHER pass: scale its launch time to cycles with this pass's clock
        cyc1 = derived["shader_clock_GHz_during_kernel"] * per_pass["pmc_sq1"]["launch_ms_in_this_pass"] * 1e6
        derived["valu_active_fraction_of_simd_cycles"] = 4 * sq["SQ_ACTIVE_INST_VALU"] / (SIMDS * cyc1)
        derived["wave_cycles_fraction_waiting"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
        derived["wave_cycles_fraction_issue_stalled"] = sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]
        derived["mean_waves_per_simd"] = 4 * sq["SQ_WAVE_CYCLES"] / (SIMDS * cyc1)
with open(os.path.join(dst, f"{tag}_sq_counters.json"), "w") as f:
    json.dump({"kernel": kern, "per_launch_average": sq, "passes": per_pass, "derived": derived,
               "selection": "dispatches of the kernel's heaviest (name, grid) group only: other grids of the same kernel name "
                            "(the 64-channel config) are not averaged in",
               "command": "rocprofv3 --pmc <8 counters> -- python3 bench.py --steps 3 --warmup 1 --no-cpu [--no-also] (two passes)"},
              f, indent=1)
# matrix-pipe utilisation of the kernels that use MFMA (north_star: "MFMA utilisation reported"): busy cycles of the matrix
# pipe over the SIMD-cycles of the dispatch (duration x shader clock x 1024 SIMDs; clock from GRBM_GUI_ACTIVE of the sq2 pass)
mf = by_name(counters("pmc_mfma"))
sq2 = by_name(counters("pmc_sq2"))
stats = {r["Name"]: r for r in csv.DictReader(open(newest(src + "/stats/*/*_kernel_stats.csv")))}
util = {}
for name, v in mf.items():
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if not busy or sum(busy) == 0:
        continue
    ns = 1e6 * sum(v["_ms"]) / len(v["_ms"])                 # this pass's own dispatch durations
    gui = v.get("GRBM_GUI_ACTIVE") or sq2.get(name, {}).get("GRBM_GUI_ACTIVE")
    clock_ghz = (sum(gui) / len(gui) / 8) / ns if gui else None
    simd_cycles = ns * clock_ghz * 1024 if clock_ghz else None
    util[name[:90]] = {"mfma_busy_cycles_per_launch": sum(busy) / len(busy),
                       "mfma_instructions_per_launch": sum(v["SQ_INSTS_MFMA"]) / len(v["SQ_INSTS_MFMA"]),
                       "avg_ms": ns / 1e6 if ns else None, "shader_clock_GHz": clock_ghz,
                       "mfma_busy_fraction_of_simd_cycles": (sum(busy) / len(busy)) / simd_cycles if simd_cycles else None}
json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA ... GRBM_GUI_ACTIVE -- python3 tools/run_path.py traffic_set 3",
           "kernels": util}, open(os.path.join(dst, f"{tag}_mfma_utilisation.json"), "w"), indent=1)
# ---- the other kernels of the path: tools/run_path.py traffic_set under FETCH_SIZE / WRITE_SIZE / SQ passes ----------------
N5 = 3 * (((1 << 22) // 3) // 256 * 256)
ALGO = [   # (kernel name part, occurrence in dispatch order, workload, algorithmic bytes per launch)
    ("k_iir_cascade_wave_pk32", 0, "config 4: 1024 ch x 8 sections x 2^20, radius 0.44 (float32 arithmetic)", 8 * 1024 * (1 << 20)),
    ("k_iir_cascade_wave_pf64w", 0, "config 4: 1024 ch x 8 sections x 2^20, radius 0.99 (double arithmetic)", 8 * 1024 * (1 << 20)),
    ("k_fir_mfma_bf16x3", 0, "config 5 float32, 2048 of its 8192 ch x 4 Mi, 1:3", (4 + 4 / 3) * 2048 * N5),
    ("k_fir_mfma_i8x", 0, "config 5 int16 bit-exact, 2048 ch x 4 Mi, 1:3", (2 + 2 / 3) * 2048 * N5),
    ("k_fir_mfma_i16", 0, "config 5 int16 fast (1 LSB), 2048 ch x 4 Mi, 1:3", (2 + 2 / 3) * 2048 * N5),
    ("k_resample_mfma_pt_f32", 0, "147:160 float32, 256 ch x 160 x 8192", (4 + 4 * 147 / 160) * 256 * 160 * 8192),
    ("k_resample_mfma_pt_f32", 1, "160:147 float32, 256 ch x 147 x 8192", (4 + 4 * 160 / 147) * 256 * 147 * 8192),
    ("k_resample_i8d", 0, "147:160 int16 bit-exact, 256 ch x 160 x 8192", (2 + 2 * 147 / 160) * 256 * 160 * 8192),
    ("k_resample_i8d", 1, "160:147 int16 bit-exact, 256 ch x 147 x 8192", (2 + 2 * 160 / 147) * 256 * 147 * 8192),
    ("k_mdct4_q15", 0, "fixed-point MDCT forward, N = 2048 x 65536 frames (int32 in, N/2 int32 out)", 6 * 2048 * 65536),
    ("k_mdct4_q15", 1, "fixed-point MDCT inverse, N = 2048 x 65536 frames", 6 * 2048 * 65536),
    ("k_mdct_reg_f32", 0, "MDCT frames analysis, F = 256, 1024 ch x 1024 frames (float32 in and out)", 8 * 256 * 1024 * 1024),
    ("k_mdct_reg_f32", 1, "MDCT frames synthesis in runs of 16 segments, same shape", 8 * 256 * 1024 * 1024),
    ("k_fir_ols8k_f32", 0, "FIR 3073 taps, 8192-point overlap-save on pairs of waves, 2048 ch x 2^20", 8 * 2048 * (1 << 20)),
]


def occurrence(agg, part, k):
    """the k-th (name, grid) group, in dispatch order, whose kernel name contains `part`"""
    groups = sorted(((min(v["_id"]), key, v) for key, v in agg.items() if part in key[0]), key=lambda g: g[0])
    return (groups[k][1], groups[k][2]) if k < len(groups) else (None, None)


def mean(x):
    return sum(x) / len(x)


def sha_of(files):
    import hashlib
    h = hashlib.sha256()
    for fn in files:
        h.update(open(os.path.join(root, "llzlab_amd", "csrc", "kernels", fn), "rb").read())
    return h.hexdigest()


SRC = {"k_iir": ["iir.hip"], "k_fir_mfma_bf16x3": ["fir_mfma.hip"], "k_fir_mfma_i16": ["fir_mfma.hip"],
       "k_fir_mfma_i8x": ["fir_mfma_i8.hip", "screen_i8.hpp"], "k_resample_mfma": ["resample_mfma.hip"],
       "k_resample_i8d": ["resample_i8.hip", "screen_i8.hpp"], "k_mdct4_q15": ["mdct_q15.hip", "fft_core.hpp"],
       "k_mdct_reg_f32": ["fft.hip", "fft_core.hpp"], "k_fir_ols8k": ["fir_ols.hip", "fft32.hpp"]}
if os.path.isdir(os.path.join(src, "set_fetch")):
    sf, sw, ssq = counters("set_fetch"), counters("set_write"), counters("set_sq")
    kernels = {}
    for part, k, what, algo in ALGO:
        (kf, vf), (kw, vw), (ks, vs) = occurrence(sf, part, k), occurrence(sw, part, k), occurrence(ssq, part, k)
        if vf is None or vw is None:
            continue
        fetch, write = 2 * mean(vf["FETCH_SIZE"]) * 1024, mean(vw["WRITE_SIZE"]) * 1024
        e = {"workload": what, "grid": int(kf[1]), "launches": len(vf["FETCH_SIZE"]),
             "fetch_bytes_x2": fetch, "write_bytes": write, "traffic_bytes_per_launch": fetch + write,
             "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic": (fetch + write) / algo,
             "launch_ms_in_fetch_pass": mean(vf["_ms"]),
             "kernel_source_sha256": sha_of(next(v for p_, v in SRC.items() if part.startswith(p_)))}
        if vs is not None and "GRBM_GUI_ACTIVE" in vs:
            cyc, ms = mean(vs["GRBM_GUI_ACTIVE"]) / 8, mean(vs["_ms"])
            e["sq"] = {"launch_ms": ms, "shader_clock_GHz": cyc / (ms * 1e6),
                       "valu_wave_instructions_per_simd": mean(vs["SQ_INSTS_VALU"]) / 1024,
                       "shader_cycles_per_valu_instruction_per_simd": cyc / (mean(vs["SQ_INSTS_VALU"]) / 1024),
                       "salu_per_valu": mean(vs["SQ_INSTS_SALU"]) / mean(vs["SQ_INSTS_VALU"]),
                       "lds_bank_conflict_fraction": (mean(vs["SQ_LDS_BANK_CONFLICT"]) / mean(vs["SQ_LDS_IDX_ACTIVE"])
                                                      if mean(vs["SQ_LDS_IDX_ACTIVE"]) else 0.0),
                       "wave_cycles_fraction_issuing": mean(vs["SQ_ACTIVE_INST_ANY"]) / mean(vs["SQ_WAVE_CYCLES"]),
                       "wave_cycles_fraction_issue_stalled": mean(vs["SQ_WAIT_INST_ANY"]) / mean(vs["SQ_WAVE_CYCLES"])}
        kernels[f"{part}#{k}"] = e
    out["kernels"] = kernels
    out["kernels_command"] = ("rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | <SQ set> -- python3 tools/run_path.py traffic_set 3 "
                              "(separate passes; FETCH_SIZE doubled as for the headline)")
    json.dump(out, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
print(json.dumps(sq, indent=1))
print(json.dumps(util, indent=1))
print(json.dumps({k: v for k, v in warm.items() if k != "all_ms"}, indent=1))
