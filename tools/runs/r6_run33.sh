#!/bin/bash
# tdf3_kernel<H>: phase offset between the two workgroups of a CU (STAGGER x 4 us for the second 256 workgroups of a launch)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $O
cd $GRAFT_REPO_ROOT
for st in 0 2 4 8 16 0 4; do
  echo "== stagger $st"
  STAGGER=$st timeout 300 tools/proto_gemm3 0 4 11 0 1 0 0 2>&1 | grep -v "amdgpu.ids" | awk '{print $1,$2,$3,$4,$5,$6,$12,$13,$14,$15,$16,$17}'
done | tee $O/tdf3h_stagger.txt
