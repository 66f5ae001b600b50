#!/bin/bash
# (1) ablation builds of tdf3_kernel on the BS-Roformer shapes: what would operands pre-split by the producer buy at most?
# (2) stress: the full-depth chunk 12 times on the bf16 x 6 kernels as the first forwards of the process
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5f
mkdir -p $O
cd $GRAFT_REPO_ROOT
for abl in 0 1 8 9 4 13; do echo "== abl $abl" >> $O/abl.log; timeout 100 tools/proto_gemm3 $abl 8 10 1 >> $O/abl.log 2>&1; done
grep "==\|rof" $O/abl.log | cut -c1-140
timeout 300 python tools/debug_rof_race.py 12 12 6first 2>&1 | grep -v amdgpu.ids | tee $O/stress.log
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k bs_roformer 2>&1 | grep "BS-Roformer\|error energy\|passed\|failed" | tee -a $O/stress.log
