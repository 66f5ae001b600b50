#!/bin/bash
# row GEMM with 16-float stages (3 workgroups per CU), finalize with the Hann table
set -u
O=gpurun_out/r2u
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "tdf or demix or finalize or full_song or sharded" 2>&1 | tail -3
for bk in 0 1; do
  ASX_TDF2_BK16=$bk timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 > $O/b_$bk.json 2>$O/b_$bk.err
  python - <<PY
import json
r=json.loads(open('$O/b_$bk.json').read().strip().splitlines()[-1])
k=r['kernel_ms']; print('BK16=$bk', r['value'], k['tdf'], k['finalize'], r['stage_roofline']['tdf']['frac'], r['stage_roofline']['finalize']['frac'], r.get('parity_rel_rms_vs_cpu'))
PY
done
ASX_TDF2_BK16=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "tdf or hq3" 2>&1 | tail -3
