#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter rows per kernel: python tools/pmc_kernel.py <dir> [substring]  (per-dispatch averages)."""
import csv
import glob
import sys
from collections import defaultdict

root, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(set)
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        if sub not in name:
            continue
        key = name.split("(")[0][:60]
        acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[key].add(row["Dispatch_Id"])
for key, counters in acc.items():
    n = max(1, len(cnt[key]))
    print(key, "dispatches", n)
    for c, v in sorted(counters.items()):
        print(f"   {c:34s} {v / n:16.1f}")
