#!/bin/bash
# tdf3_kernel<H>: eight waves sharing one x tile (128 x 384 / 128 x 256 column tiles, one workgroup per CU) against the 4-wave 128 x 192 / 128 x 128 forms
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $O
cd $GRAFT_REPO_ROOT
for t in 0 3 1 4 0 3; do for m in 0 1; do
  echo "== tile $t map $m"
  timeout 300 tools/proto_gemm3 0 4 11 $m 1 0 $t 2>&1 | grep -v "amdgpu.ids" | awk '{print $1,$2,$3,$4,$5,$6,$12,$13,$14,$15,$16,$17, "relrms", $(NF-13)}'
done; done | tee $O/tdf3h_eight_waves.txt
