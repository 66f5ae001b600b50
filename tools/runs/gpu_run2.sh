#!/bin/bash
# round-2 GPU run 2: full GPU suite (no -x), TDF ablations, tile-shape A/B, per-dispatch trace of the row GEMMs
set -u
O=gpurun_out/r2b
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
B="python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --siblings 0"
ASX_TDF2=0 timeout 300 $B > $O/b_v1.json 2> $O/b_v1.err
for abl in 0 1 4 5 2 3 7; do
  ASX_TDF2=1 ASX_TDF2_ABL=$abl timeout 300 $B > $O/b_v2_abl$abl.json 2> $O/b_v2_abl$abl.err
done
ASX_TDF2=0 ASX_GEMM_T128=2 timeout 300 $B > $O/b_v1_t128.json 2> $O/b_v1_t128.err
ASX_TDF2=1 ASX_GEMM_T128=2 timeout 300 $B > $O/b_v2_t128.json 2> $O/b_v2_t128.err
cd /tmp && export TMPDIR=/tmp
ASX_TDF2=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_v2 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --siblings 0 > $GRAFT_REPO_ROOT/$O/trace_v2.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r2b/b_*.json')):
    try:
        r=json.load(open(f)); print(os.path.basename(f), r['value'], r['ms_per_step'], {k:r['kernel_ms'][k] for k in ('tdf','down','up','conv3x3')}, r['stage_roofline']['tdf']['frac'])
    except Exception as e: print(f,'ERR',e)
PY
tail -4 $O/pytest_gpu.log
ls $O/trace_v2 | head
