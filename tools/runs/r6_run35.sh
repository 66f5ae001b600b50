#!/bin/bash
# BS-Roformer sibling with the feed-forward's hidden activations as a pair image (on / off), no CPU leg
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  ASX_PAIR_IMAGES=$v timeout 900 python tools/bench_siblings.py --workloads roformer --cpu 0 --steps 2 --warmup 1 2>/dev/null | tail -1 > $O/rof_pair_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r6f/rof_pair_$v.json"))
print("pair images $v:", d["value"], d["ms_per_step"], {k:v for k,v in d.get("kernel_ms",{}).items()})
PY
done | tee $O/rof_pair_ab.txt
