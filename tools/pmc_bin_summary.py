#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_bin.sh: python tools/pmc_bin_summary.py <outdir> [kernel substring]"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:36s} n={len(v):3d} last={v[-1]:.6g} mean={sum(v) / len(v):.6g}")
