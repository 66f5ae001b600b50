"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (sum over dispatches)."""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(float)
for f in glob.glob(root + "/*/p_counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void asx::", "").replace("asx::", "")[:64]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
        key = (f, r["Dispatch_Id"])
        if key not in seen and "FETCH" in f:
            seen.add(key)
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
for k in sorted(agg, key=lambda k: -dur.get(k, 0)):
    print(k, f"  [{dur.get(k,0)*1e3:.2f} ms in the FETCH pass]")
    for c, v in sorted(agg[k].items()):
        print(f"    {c:34s} total {v:18.1f}  per-dispatch {v / max(1, cnt[k][c]):16.1f}  n={cnt[k][c]}")
