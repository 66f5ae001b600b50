#!/bin/bash
# BS-Roformer sibling: the feed-forward's GELU on libm erff (2, default), on fast_erf (6: Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7), as ReLU (1: timing probe, wrong results)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6g
mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in 2 6 1 2 6; do
  ASX_ROF_GELU=$v timeout 900 python tools/bench_siblings.py --workloads roformer --cpu 0 --steps 2 --warmup 1 2>/dev/null | tail -1 > $O/rof_gelu_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r6g/rof_gelu_$v.json"))
print("gelu $v:", d["value"], d["ms_per_step"], {k:v for k,v in d.get("kernel_ms",{}).items() if 'GEMM' in k or 'atten' in k})
PY
done | tee $O/rof_gelu_ab.txt
