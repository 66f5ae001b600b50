#!/bin/bash
set -u
O=gpurun_out/r3d
mkdir -p $O
for ip in 0 1 0 1; do
  ASX_TDF_INPLACE=$ip timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 > $O/b_$ip.json 2>$O/b_$ip.err
  python - <<PY
import json
r=json.loads(open('$O/b_$ip.json').read().strip().splitlines()[-1])
k=r['kernel_ms']; print('INPLACE=$ip', r['value'], k['conv3x3'], k['tdf'], r.get('parity_rel_rms_vs_cpu'))
PY
done
