#!/bin/bash
mkdir -p gpurun_out/r6a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=tools/proto_conv3h
{
  timeout 300 $P 0 2 55 5
  for abl in 2 4 8 6; do timeout 120 $P $abl 2 55 5 1 | tail -2 | head -1; done
  timeout 120 $P 32 2 55 3 1 | grep -E 'slot|step (1[0-9]):|level 0'
} > gpurun_out/r6a/conv3h_run12.txt 2>&1
cat gpurun_out/r6a/conv3h_run12.txt | cut -c1-260
