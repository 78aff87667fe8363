#!/usr/bin/env python3
"""Summarise a tools/pmc.sh output directory: per-kernel average duration and counters."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(d + "/stats/*/*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if filt in r["Name"]:
            print("stats", r["Name"][:60], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3))
for f in sorted(glob.glob(d + "/pmc*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"] and "synth" not in r["Kernel_Name"] and "tail" not in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print("%-42s %-22s n=%d avg=%.5g" % (k[0], k[1], len(v), sum(v) / len(v)))
