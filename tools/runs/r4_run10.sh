#!/bin/bash
# bf16x6 row GEMM prototype: correctness vs float64 and the fp32-MFMA kernel, time per launch; then the ablation builds on the TDF shapes
set -u
O=gpurun_out/r4j
mkdir -p $O
timeout 300 tools/proto_gemm3 0 > $O/proto_all.txt 2>&1; echo "rc=$?" >> $O/proto_all.txt
cat $O/proto_all.txt
for abl in 1 2 4 8 9 13; do
  timeout 120 tools/proto_gemm3 $abl 2 3 > $O/proto_abl$abl.txt 2>&1
  echo "abl $abl"; cut -c1-150 $O/proto_abl$abl.txt
done
