#!/bin/bash
# what bounds tdf3_kernel<3, 8, H>: ablations of the fp16 x 3 row GEMM on the TDF / Roformer shapes (1 no split, 4 no epilogue traffic, 8 no W loads, 16 no rescale)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $O
cd $GRAFT_REPO_ROOT
for abl in 0 1 16 17 4 8 21 29; do
  echo "== abl $abl"
  timeout 300 tools/proto_gemm3 $abl 4 10 0 1 0 0 2>&1 | grep -v "amdgpu.ids" | awk '{print $1,$2,$3,$4,$5,$6,$12,$13,$14,$15,$16,$17}'
done | tee $O/tdf3h_abl.txt
