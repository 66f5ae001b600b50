#!/bin/bash
# round-2 GPU run 4: small-tile A/B for the short-K row GEMMs, then the one-off full-song parity (CPU oracle: ~15 min)
set -u
O=gpurun_out/r2d
mkdir -p $O
B="python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --siblings 0"
ASX_TDF2_SMALL=768 timeout 300 $B > $O/b_small768.json 2> $O/b_small768.err
ASX_TDF2_SMALL=200 timeout 300 $B > $O/b_small200.json 2> $O/b_small200.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r2d/b_*.json')):
    try:
        r=json.load(open(f)); km=r['kernel_ms']; print(os.path.basename(f), r['value'], r['ms_per_step'], km.get('tdf'))
    except Exception as e: print(f,'ERR',e)
PY
timeout 2000 python tools/fullsong_parity.py > $O/fullsong_parity.json 2> $O/fullsong_parity.err
tail -5 $O/fullsong_parity.err; cat $O/fullsong_parity.json
