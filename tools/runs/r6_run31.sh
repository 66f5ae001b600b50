#!/bin/bash
# tdf3_kernel<H>: what the split costs -- abl 64 = pair-image reader on fp32 data (loads + LDS stores stay, split arithmetic gone), abl 1 = no loads either
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $O
cd $GRAFT_REPO_ROOT
for abl in 0 64 1 0 64; do
  echo "== abl $abl"
  timeout 300 tools/proto_gemm3 $abl 4 10 0 1 0 0 2>&1 | grep -v "amdgpu.ids" | awk '{print $1,$2,$3,$4,$5,$6,$12,$13,$14,$15,$16,$17}'
done | tee $O/tdf3h_split_cost.txt
