#!/bin/bash
# fp16 x 3 row GEMM with per-ROW exponents: rescale of the accumulators as a rare branch (tools/proto_gemm3) against a branch-free
# multiply in front of every row group's MFMAs (tools/proto_gemm3_u, -DASX_H_UNCOND), one call
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5o
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 tools/proto_gemm3 0 3 99 0 1 > $O/gemm3_f16x3_rows_branch.txt 2>&1; echo "rc $?" >> $O/gemm3_f16x3_rows_branch.txt
timeout 300 tools/proto_gemm3_u 0 3 99 0 1 > $O/gemm3_f16x3_rows_uncond.txt 2>&1; echo "rc $?" >> $O/gemm3_f16x3_rows_uncond.txt
cat $O/gemm3_f16x3_rows_branch.txt $O/gemm3_f16x3_rows_uncond.txt | cut -c1-60,100-140,175-400
