"""print the essentials of a bench.py line: python tools/show_bench_line.py [file] (default gpurun_out/r2/bench_4rank.json)"""
import sys
import json
d=json.loads([l for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r2/bench_4rank.json") if l.startswith("{")][0])
print(d["value"], d["n_gpus"], d["table_broadcast"][:60])
print({k:(round(v["ms"],3), v.get("channels_per_gpu")) for k,v in d["also"].items()})
