#!/bin/bash
# tdf3_kernel<H>: the accumulator rescale behind a workgroup-uniform stage stamp (abl 0) against the unconditional multiply (abl 32); correctness on the small / spread shapes
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $O
cd $GRAFT_REPO_ROOT
for abl in 0 32 0 32; do
  echo "== abl $abl"
  timeout 300 tools/proto_gemm3 $abl 4 10 0 1 0 0 2>&1 | grep -v "amdgpu.ids" | awk '{print $1,$2,$3,$4,$5,$6,$12,$13,$14,$15,$16,$17}'
done | tee $O/tdf3h_rescale_ab.txt
echo "== correctness (abl 0), all columns" | tee -a $O/tdf3h_rescale_ab.txt
timeout 300 tools/proto_gemm3 0 0 3 0 1 0 0 2>&1 | grep -v "amdgpu.ids" | tee -a $O/tdf3h_rescale_ab.txt
timeout 300 tools/proto_gemm3 0 16 18 0 1 1 0 2>&1 | grep -v "amdgpu.ids" | tee -a $O/tdf3h_rescale_ab.txt
timeout 300 tools/proto_gemm3 0 12 15 0 1 1 0 2>&1 | grep -v "amdgpu.ids" | tee -a $O/tdf3h_rescale_ab.txt
