#!/bin/bash
set -u
O=gpurun_out/r2h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_roformer.py -m gpu -q -x > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log
tail -3 $O/pytest_parity.log
B="python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --siblings 0"
timeout 300 $B > $O/b_new.json 2> $O/b_new.err
ASX_TDF2=0 timeout 300 $B > $O/b_v1.json 2> $O/b_v1.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r2h/b_*.json')):
    try:
        r=json.load(open(f)); km=r['kernel_ms']; print(os.path.basename(f), r['value'], r['ms_per_step'], {k:km.get(k) for k in ('tdf','up','down','conv3x3','istft')}, r['stage_roofline']['tdf']['frac'])
    except Exception as e: print(f,'ERR',e)
PY
ASX_TDF2=1 ASX_TDF2_ABL=16 timeout 300 python tools/probe_tdf_timeline.py 2> $O/timeline.err | head -8
