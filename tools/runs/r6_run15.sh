#!/bin/bash
mkdir -p gpurun_out/r6a
cd $GRAFT_REPO_ROOT
for abl in 34 36 38 40; do echo "abl $abl (32 timeline; +2 no loads, +4 no stores, +8 no split)"; timeout 120 tools/proto_conv3h $abl 2 55 3 1 | grep -E 'step (1[0-3]):|level 0'; done > gpurun_out/r6a/conv3h_abl3.txt 2>&1
cat gpurun_out/r6a/conv3h_abl3.txt
