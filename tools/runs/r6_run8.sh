#!/bin/bash
# round 6, run 8: conv3h_kernel with three register sets (loads first in the step): correctness, time, ablations, timeline
mkdir -p gpurun_out/r6a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=tools/proto_conv3h
{
  timeout 300 $P 0 2 55 5
  for abl in 2 4 8 6; do timeout 120 $P $abl 2 55 5 1 | tail -2 | head -1; done
  timeout 120 $P 32 2 55 3 1 | head -28
} > gpurun_out/r6a/conv3h_run8.txt 2>&1
tail -50 gpurun_out/r6a/conv3h_run8.txt
