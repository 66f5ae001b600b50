#!/bin/bash
# down conv with two-channel stages: parity + A/B
set -u
O=gpurun_out/r2t
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
for kc2 in 1 0; do
  ASX_DOWN_KC2=$kc2 timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 > $O/b_$kc2.json 2>$O/b_$kc2.err
  python - <<PY
import json
r=json.loads(open('$O/b_$kc2.json').read().strip().splitlines()[-1])
k=r['kernel_ms']; print('KC2=$kc2', r['value'], k['down'], r['stage_roofline']['down'])
PY
done
