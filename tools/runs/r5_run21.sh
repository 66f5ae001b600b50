#!/bin/bash
# tile -> XCD maps of the row GEMM again, now on the fp16 x 3 arithmetic (half the MFMA time: the memory side weighs more), tile forms 0 / 1
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5u
mkdir -p $O
cd $GRAFT_REPO_ROOT
for t in 0 1; do for m in 0 1 2; do
  echo "== tile $t map $m"
  timeout 300 tools/proto_gemm3 0 3 11 $m 1 0 $t 2>&1 | grep -v "amdgpu.ids" | awk '{print $1,$2,$3,$4,$5,$6,$12,$13,$14,$15,$16,$17}'
done; done | tee $O/f16x3_tile_map.txt
