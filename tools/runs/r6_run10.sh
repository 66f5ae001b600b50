#!/bin/bash
mkdir -p gpurun_out/r6a
cd $GRAFT_REPO_ROOT
{ timeout 120 tools/proto_conv3h 64 2 55 5 1 | tail -2; timeout 120 tools/proto_conv3h 96 2 55 3 1 | sed -n 5,12p; } > gpurun_out/r6a/conv3h_prio.txt 2>&1
cat gpurun_out/r6a/conv3h_prio.txt
