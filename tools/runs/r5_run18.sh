#!/bin/bash
# which BS-Roformer GEMM makes the fp16 x 3 leg non-deterministic?  ASX_F16X3_N restricts the arithmetic to the launches with that N
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5r
mkdir -p $O
cd $GRAFT_REPO_ROOT
for n in "$@"; do
  echo "== ASX_F16X3_N=$n"; ASX_F16X3_N=$n timeout 300 python tools/debug_rof_race.py 12 3 2>&1 | grep -v "amdgpu.ids"
done | tee $O/rof_bisect.txt
