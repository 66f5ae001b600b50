#!/bin/bash
# round 6, run 4: is conv3h_kernel DRAM-bound or CU-bound?  Batch items aliased onto one 151-MB plane set (reads, then writes too)
mkdir -p gpurun_out/r6a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=tools/proto_conv3h
{
  for al in 0 1 2; do for abl in 0 2 4; do echo "alias $al abl $abl"; timeout 120 $P $abl 2 55 5 1 $al | tail -2 | head -1; done; done
} > gpurun_out/r6a/conv3h_run4.txt 2>&1
cat gpurun_out/r6a/conv3h_run4.txt
