#!/bin/bash
# race hunt: bf16 x 6 leg of the BS-Roformer chunk (nondeterministic 3e-5 .. 7e-5 on two boxes, 1.5e-6 on a third)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5d
mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== all bf16x6 kernels, depth 12" | tee -a $O/race.log; timeout 200 python tools/debug_rof_race.py 12 4 2>&1 | grep -v amdgpu.ids | tee -a $O/race.log
echo "== ASX_ATTN6=0 (row GEMM only), depth 12" | tee -a $O/race.log; ASX_ATTN6=0 timeout 200 python tools/debug_rof_race.py 12 4 2>&1 | grep -v amdgpu.ids | tee -a $O/race.log
echo "== all, depth 1" | tee -a $O/race.log; timeout 200 python tools/debug_rof_race.py 1 4 2>&1 | grep -v amdgpu.ids | tee -a $O/race.log
echo "== ASX_ATTN6_QW=1 (64-query workgroups), depth 12" | tee -a $O/race.log; ASX_ATTN6_QW=1 timeout 200 python tools/debug_rof_race.py 12 3 2>&1 | grep -v amdgpu.ids | tee -a $O/race.log
echo "== ASX_TDF3_MAP=0, depth 12" | tee -a $O/race.log; ASX_TDF3_MAP=0 timeout 200 python tools/debug_rof_race.py 12 3 2>&1 | grep -v amdgpu.ids | tee -a $O/race.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "variants or groupnorm or winograd_bf16x6 or abi" 2>&1 | tail -5 | tee -a $O/race.log
