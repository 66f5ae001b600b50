#!/bin/bash
# batch-size knobs of the sibling loops on the final kernels (round 3 found them worth more than kernel changes): chunks / segments /
# patches per net pass
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5k
mkdir -p $O
cd $GRAFT_REPO_ROOT
for mb in 8 16 31; do echo "== roformer max_batch $mb" | tee -a $O/batch.log; timeout 200 python tools/probe_roformer.py 241 $mb 2>&1 | grep "audio\|gemm \|attention" | tee -a $O/batch.log; done
for mb in 28 42 56 84; do echo "== htdemucs max_batch $mb" | tee -a $O/batch.log; timeout 200 python tools/probe_demucs.py 240 $mb 2 2>&1 | grep -i "wall\|rtf" | head -3 | tee -a $O/batch.log; done
for mb in 32 48 84; do echo "== vr max_batch $mb" | tee -a $O/batch.log; timeout 200 python tools/probe_vr.py 240 $mb 2>&1 | grep -i "wall\|rtf" | head -3 | tee -a $O/batch.log; done
