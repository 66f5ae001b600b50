#!/bin/bash
set -u
O=gpurun_out/r2j
mkdir -p $O
B="python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --siblings 0"
for k in 144 192 240; do ASX_CONV_KC4=$k timeout 300 $B > $O/b_kc4_$k.json 2> $O/b_kc4_$k.err; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r2j/b_*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); km=r['kernel_ms']; print(os.path.basename(f), r['value'], r['ms_per_step'], {k:km.get(k) for k in ('conv3x3','tdf')}, r['roofline']['frac'])
    except Exception as e: print(f,'ERR',e)
PY
