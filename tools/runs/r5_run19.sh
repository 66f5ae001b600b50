#!/bin/bash
# the BS-Roformer qkv projection (rotary epilogue + per-row factor) on the harness: whole-output determinism, which epilogue feature,
# inline-asm descale (proto_gemm3) against plain C (proto_gemm3_u)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5s
mkdir -p $O
cd $GRAFT_REPO_ROOT
for b in proto_gemm3 proto_gemm3_u; do
  echo "== $b"
  timeout 600 tools/$b 0 12 15 0 1 1 0 2>&1 | grep -v "amdgpu.ids" | cut -c1-200
done | tee $O/qkv_rot.txt
