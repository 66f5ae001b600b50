#!/bin/bash
# the bf16 x 6 leg as the FIRST forward after a load, plain and with poisoned allocations (ASX_POISON)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5e
mkdir -p $O
cd $GRAFT_REPO_ROOT
for cfg in "none 12 6first" "255 12 6first" "127 12 6first" "255 12 32first" "255 1 6first"; do
  set -- $cfg
  echo "== ASX_POISON=$1 depth $2 $3" | tee -a $O/poison.log
  if [ $1 = none ]; then unset ASX_POISON; else export ASX_POISON=$1; fi
  timeout 200 python tools/debug_rof_race.py $2 3 $3 2>&1 | grep -v amdgpu.ids | tee -a $O/poison.log
done
export ASX_POISON=255
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsong.py 2>&1 | tail -15 | tee -a $O/poison.log
