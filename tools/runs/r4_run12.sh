#!/bin/bash
# bf16x6 Winograd prototype: correctness on small shapes, time per launch per level, ablation builds on L0 / L1
set -u
O=gpurun_out/r4l
mkdir -p $O
timeout 300 tools/proto_wino6 0 > $O/proto_all.txt 2>&1; echo "rc=$?" >> $O/proto_all.txt
cat $O/proto_all.txt
for abl in 1 2 4 8 16 15; do
  timeout 120 tools/proto_wino6 $abl 4 5 > $O/proto_abl$abl.txt 2>&1
  echo "abl $abl"; cut -c1-160 $O/proto_abl$abl.txt
done
