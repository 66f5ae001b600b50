#!/bin/bash
# round-2 GPU run 5: fast-FFT refinements (single exchange buffer, Hann table, G sweep), small-tile A/B, full-song parity
set -u
O=gpurun_out/r2e
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log
tail -3 $O/pytest_parity.log
B="python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --siblings 0"
for g in 16 32 8 12; do ASX_FFT3_G=$g timeout 300 $B > $O/b_g$g.json 2> $O/b_g$g.err; done
ASX_TDF2_SMALL=768 timeout 300 $B > $O/b_small768.json 2> $O/b_small768.err
ASX_TDF2_SMALL=200 timeout 300 $B > $O/b_small200.json 2> $O/b_small200.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r2e/b_*.json')):
    try:
        r=json.load(open(f)); km=r['kernel_ms']; print(os.path.basename(f), r['value'], r['ms_per_step'], {k:km.get(k) for k in ('stft','istft','finalize','tdf')}, {k:r['stage_roofline'][k]['frac'] for k in ('stft','istft') if k in r['stage_roofline']})
    except Exception as e: print(f,'ERR',e)
PY
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 2000 python tools/fullsong_parity.py > $O/fullsong_parity.json 2> $O/fullsong_parity.err
tail -4 $O/fullsong_parity.err; cat $O/fullsong_parity.json
