#!/bin/bash
# round 6, run 18: conv3h_kernel with the generalised walk (bands of 32 / 16 / 8 strips) and the accumulate mode: level 1 (96 -> 96) as four launches
mkdir -p gpurun_out/r6b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=tools/proto_conv3h
{
  timeout 300 $P 0 2 55 5 0 0 32
  timeout 300 $P 0 2 55 0 0 0 16 | grep -v 'level 0'
  timeout 300 $P 0 2 55 0 0 0 8 | grep -v 'level 0'
  for bw in 32 16 8; do timeout 300 $P 0 2 55 5 2 0 $bw; done
} > gpurun_out/r6b/conv3h_l1.txt 2>&1
cat gpurun_out/r6b/conv3h_l1.txt
