#!/bin/bash
# whole-output determinism of the fp16 x 3 row GEMM on the harness shapes (two runs bit for bit, and against the fp32-MFMA kernel),
# per tile form: $1 = 0 (128 x 192) / 1 (128 x 128) / 2 (64 x 128)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5q
mkdir -p $O
cd $GRAFT_REPO_ROOT
for t in "$@"; do
  echo "== tile $t"
  timeout 600 tools/proto_gemm3 0 0 99 0 1 1 $t 2>&1 | grep -v "amdgpu.ids" | cut -c1-150
done | tee $O/gemm3_f16x3_full_tiles.txt
