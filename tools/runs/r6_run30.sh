#!/bin/bash
# a TDF block's two linears with the bottleneck activations as a pair image (gemm1 writes the two fp16 parts, gemm2 multiplies them as they are)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 tools/proto_gemm3 0 0 0 0 1 0 0 1 2>&1 | grep -v "amdgpu.ids" | tee $O/tdf_pair_image_chain.txt
