#!/bin/bash
set -u
O=gpurun_out/r2f
mkdir -p $O
ASX_TDF2=1 ASX_TDF2_ABL=16 timeout 300 python tools/probe_tdf_timeline.py > $O/timeline_m1.txt 2> $O/timeline_m1.err
ASX_TDF2=2 ASX_TDF2_ABL=16 timeout 300 python tools/probe_tdf_timeline.py > $O/timeline_m2.txt 2> $O/timeline_m2.err
cat $O/timeline_m1.txt; tail -3 $O/timeline_m1.err
B="python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --siblings 0"
for g in 16 32; do ASX_FFT3_G=$g timeout 300 $B > $O/b_g$g.json 2> $O/b_g$g.err; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r2f/b_*.json')):
    try:
        r=json.load(open(f)); km=r['kernel_ms']; print(os.path.basename(f), r['value'], r['ms_per_step'], {k:km.get(k) for k in ('stft','istft','finalize','tdf')}, {k:r['stage_roofline'][k]['frac'] for k in ('stft','istft') if k in r['stage_roofline']})
    except Exception as e: print(f,'ERR',e)
PY
