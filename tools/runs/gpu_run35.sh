#!/bin/bash
set -u
O=gpurun_out/r3c
mkdir -p $O
for mb in 1 2 4 0; do
  timeout 300 python bench.py --steps 4 --warmup 1 --cpu-seconds 0 --siblings 0 --max-batch $mb > $O/b_$mb.json 2>$O/b_$mb.err
  python - <<PY
import json
r=json.loads(open('$O/b_$mb.json').read().strip().splitlines()[-1])
k=r['kernel_ms']; print('max_batch=$mb', r['value'], r['ms_per_step'], k['conv3x3'], k['tdf'], k['down'], k['up'], k['conv1x1'])
PY
done
