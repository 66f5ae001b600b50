#!/bin/bash
set -u
O=gpurun_out/r2k
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log
tail -2 $O/pytest_parity.log
ASX_TDF2_BK16=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_roformer.py -m gpu -q -x > $O/pytest_bk16.log 2>&1; echo "rc=$?" >> $O/pytest_bk16.log
tail -2 $O/pytest_bk16.log
B="python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --siblings 0"
timeout 300 $B > $O/b_base.json 2> $O/b_base.err
ASX_TDF2_BK16=1 timeout 300 $B > $O/b_bk16_1.json 2> $O/b_bk16_1.err
ASX_TDF2_BK16=2 timeout 300 $B > $O/b_bk16_2.json 2> $O/b_bk16_2.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r2k/b_*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); km=r['kernel_ms']; print(os.path.basename(f), r['value'], r['ms_per_step'], {k:km.get(k) for k in ('conv3x3','tdf')}, r['roofline']['frac'], r['stage_roofline']['tdf']['frac'])
    except Exception as e: print(f,'ERR',e)
PY
