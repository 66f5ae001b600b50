#!/bin/bash
# non-temporal epilogue stores / residual loads: A/B
set -u
O=gpurun_out/r3b
mkdir -p $O
for nt in 0 1 2 6 7; do
  ASX_NT=$nt timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 > $O/b_$nt.json 2>$O/b_$nt.err
  python - <<PY
import json
r=json.loads(open('$O/b_$nt.json').read().strip().splitlines()[-1])
k=r['kernel_ms']; print('NT=$nt', r['value'], k['conv3x3'], k['tdf'], k['down'], k['up'])
PY
done
