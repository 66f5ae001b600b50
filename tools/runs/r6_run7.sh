#!/bin/bash
# round 6, run 7: per-step timeline of one workgroup of conv3h_kernel (s_memtime stamps)
mkdir -p gpurun_out/r6a
cd $GRAFT_REPO_ROOT
timeout 120 tools/proto_conv3h 32 2 55 3 1 > gpurun_out/r6a/conv3h_timeline.txt 2>&1
cat gpurun_out/r6a/conv3h_timeline.txt
