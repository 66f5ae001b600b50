#!/bin/bash
# istft3p timing probes (ABL: results invalid) + parity of the product configuration
set -u
O=gpurun_out/r2r
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
for abl in 0 1 2 3; do
  ASX_ISTFT_ABL=$abl timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 > $O/b_$abl.json 2>$O/b_$abl.err
  python - <<PY
import json
r=json.loads(open('$O/b_$abl.json').read().strip().splitlines()[-1])
k=r['kernel_ms']; print('ABL=$abl', r['value'], {x:k[x] for x in k if 'stft' in x}, r['stage_roofline']['stft']['frac'], r['stage_roofline']['istft']['frac'])
PY
done
