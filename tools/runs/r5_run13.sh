#!/bin/bash
# fp16 x 3 row GEMM (tdf3_kernel<..., H = true>) against the bf16 x 6 form on the stand-alone harness, one call: time per launch and
# distance to a float64 GEMM on the TDF / Roformer / Demucs shapes and on the dynamic-range ("spread") shapes
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5m
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 tools/proto_gemm3 0 0 99 0 0 > $O/gemm3_bf16x6.txt 2>&1; echo "rc $?" >> $O/gemm3_bf16x6.txt
timeout 300 tools/proto_gemm3 0 0 99 0 1 > $O/gemm3_f16x3.txt 2>&1; echo "rc $?" >> $O/gemm3_f16x3.txt
cat $O/gemm3_bf16x6.txt $O/gemm3_f16x3.txt
