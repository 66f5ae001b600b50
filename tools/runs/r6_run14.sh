#!/bin/bash
mkdir -p gpurun_out/r6a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=tools/proto_conv3h
{
  timeout 300 $P 0 2 55 5
  timeout 120 $P 1024 2 55 5 1 | tail -2 | head -1
  timeout 120 $P 32 2 55 3 1 | grep -E 'slot 1|step (1[0-5]):|level 0'
} > gpurun_out/r6a/conv3h_run14.txt 2>&1
cat gpurun_out/r6a/conv3h_run14.txt | cut -c1-200
