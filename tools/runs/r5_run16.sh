#!/bin/bash
# BS-Roformer depth 12, one chunk: is the split-operand leg deterministic, and is the FIRST forward after the load the odd one?
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5p
mkdir -p $O
cd $GRAFT_REPO_ROOT
( echo "== f16x3, fp32 leg first"; timeout 300 python tools/debug_rof_race.py 12 4
  echo "== f16x3, split leg first"; timeout 300 python tools/debug_rof_race.py 12 4 6first
  echo "== bf16x6, split leg first"; ASX_GEMM_F16X3=0 timeout 300 python tools/debug_rof_race.py 12 4 6first
  echo "== f16x3, attention on the fp32 pipe, split leg first"; ASX_ATTN6=0 timeout 300 python tools/debug_rof_race.py 12 4 6first
) 2>&1 | grep -v "amdgpu.ids" | tee $O/rof_race.txt
