#!/bin/bash
mkdir -p gpurun_out/r6a
cd $GRAFT_REPO_ROOT
timeout 120 tools/proto_conv3h 32 2 55 3 1 > gpurun_out/r6a/conv3h_timeline2.txt 2>&1
head -8 gpurun_out/r6a/conv3h_timeline2.txt | cut -c1-700; tail -3 gpurun_out/r6a/conv3h_timeline2.txt
