#!/bin/bash
# round 6, run 25: TDF tile A/B -- 64 x 128 tiles (more workgroups per CU, more x loads in flight) on all / long-K layers
mkdir -p gpurun_out/r6e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 4096 512; do
ASX_TDF2_SMALL=$v timeout 600 python bench.py --steps 5 --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored --no-arith-ab 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ASX_TDF2_SMALL=$v', d['value'], d['kernel_ms']['tdf'], {k:v['avg_launch_ms'] for k,v in d['roofline']['per_level']['tdf'].items() if k.startswith('L0') or k.startswith('L1')})"
done > gpurun_out/r6e/tdf_small_ab.txt 2>&1
cat gpurun_out/r6e/tdf_small_ab.txt
