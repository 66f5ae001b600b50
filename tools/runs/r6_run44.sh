#!/bin/bash
# conv_down6_kernel: tests again, per-level launch times (rocprofv3 kernel trace of one bench run)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6h
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "down_conv" 2>&1 | grep -v "^$" | tail -12 | tee $O/pytest_down6.txt
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored --no-arith-ab > /dev/null 2>&1)
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r6h/trace/**/t_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "down6" in r["Kernel_Name"] or "ConvDmaCfg<1, 1, 1, 0, 6, 8, 1, 1>" in r["Kernel_Name"]]
by = collections.defaultdict(list)
for r in rows:
    by[(r["Kernel_Name"][:60], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(k, len(v), "avg ms %.3f" % (sum(v) / len(v)))
PY
