#!/bin/bash
# round 6, run 1: first contact of conv3h_kernel (direct fp16 x 3 conv, level 0) -- correctness on the small shapes, then time per
# launch at B = 55 for the three tile orders and the ablation builds
mkdir -p gpurun_out/r6a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=tools/proto_conv3h
{
  timeout 300 $P 0 0 55 5
  timeout 120 $P 0 1 55 5 | tail -3
  timeout 120 $P 0 2 55 5 | tail -3
  for abl in 1 2 4 8 6 7 14; do timeout 120 $P $abl 2 55 5 | tail -2; done
} > gpurun_out/r6a/conv3h.txt 2>&1
tail -40 gpurun_out/r6a/conv3h.txt
