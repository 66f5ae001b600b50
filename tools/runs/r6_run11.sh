#!/bin/bash
mkdir -p gpurun_out/r6a
cd $GRAFT_REPO_ROOT
for abl in 160 288 416; do echo "abl $abl (32 timeline + 128 stores at the tile's end + 256 no rescale)"; timeout 120 tools/proto_conv3h $abl 2 55 3 1 | grep -E 'step (1[0-5]):|level 0'; done > gpurun_out/r6a/conv3h_abl2.txt 2>&1
cat gpurun_out/r6a/conv3h_abl2.txt
