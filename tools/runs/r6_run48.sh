#!/bin/bash
# per-level launch times of conv_down6_kernel / conv_up6_kernel (rocprofv3 kernel trace of a short bench run) against their HBM floors
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6h
mkdir -p $O; rm -rf $O/trace2
cd $GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace2 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored --no-arith-ab > /dev/null 2>&1)
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r6h/trace2/**/t_kernel_trace.csv", recursive=True)[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "down6" in n or "up6" in n:
        by[(n[:44], int(r["Grid_Size"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
# HQ_3 geometry, 55 chunks: level l tensor = 48 (l + 1) channels x (256 >> l) x (3072 >> l) floats per chunk
def gb(l): return 55 * 48 * (l + 1) * (256 >> l) * (3072 >> l) * 4 / 1e9
lv_down = {}; lv_up = {}
for (n, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
    print(f"{n:44s} grid {g:9d}  launches {len(v):2d}  avg {sum(v) / len(v):7.3f} ms")
print("HBM bytes: down l -> l+1 = x_l + y_(l+1); up l+1 -> l = x_(l+1) + skip_l + y_l")
for l in range(5):
    print(f"  level {l}: down {gb(l) + gb(l + 1):6.2f} GB  up {gb(l + 1) + 2 * gb(l):6.2f} GB")
PY
