#!/bin/bash
# HBM traffic of the level-change kernels per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each, kernel trace only) around a short bench run
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6h/pmc_updown
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
for grp in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$grp -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored --no-arith-ab > $O/$grp.log 2>&1)
done
python3 - <<'PY' | tee gpurun_out/r6h/pmc_updown.txt
import csv, glob, collections
root = "gpurun_out/r6h/pmc_updown"
val = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    tr = {}
    for f in glob.glob(f"{root}/{grp}/**/p_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            tr[r["Dispatch_Id"]] = int(r["Grid_Size_X"])
    for f in glob.glob(f"{root}/{grp}/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if ("down6" in n or "up6" in n) and r["Counter_Name"] == grp:
                key = (n.replace("void asx::", "")[:28], tr.get(r["Dispatch_Id"], int(r.get("Grid_Size", 0) or 0)))
                val[key][grp].append(float(r["Counter_Value"]))
def gb(l): return 55 * 48 * (l + 1) * (256 >> l) * (3072 >> l) * 4 / 1e9
alg = {}
for l in range(5):
    alg[("down", l)] = gb(l) + gb(l + 1)
    alg[("up", l)] = gb(l + 1) + 2 * gb(l)
print("# FETCH_SIZE x 2 (gfx950 note of MI355X_MICROARCH.md) and WRITE_SIZE, KiB -> GB, per launch (mean over the launches of the run); levels by grid size, largest first")
for kind in ("down6", "up6"):
    keys = sorted([k for k in val if kind in k[0]], key=lambda k: -k[1])
    for l, k in enumerate(keys):
        f = sum(val[k]["FETCH_SIZE"]) / max(1, len(val[k]["FETCH_SIZE"])) * 1024 * 2 / 1e9
        w = sum(val[k]["WRITE_SIZE"]) / max(1, len(val[k]["WRITE_SIZE"])) * 1024 / 1e9
        a = alg[("down" if kind == "down6" else "up", l)]
        print(f"{k[0]:28s} level {l}: fetch {f:6.2f} GB + write {w:6.2f} GB = {f + w:6.2f} GB; algorithmic {a:6.2f} GB; ratio {(f + w) / a:.2f}")
PY
