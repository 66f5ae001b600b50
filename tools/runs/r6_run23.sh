#!/bin/bash
# round 6, run 23: conv3h_kernel fragment reads two stages ahead (three register sets) against one (ABL 2048), timeline of both
mkdir -p gpurun_out/r6d
cd $GRAFT_REPO_ROOT
P=tools/proto_conv3h
{
  for i in 1 2; do
    timeout 120 $P 0 2 55 5 1 | tail -2 | head -1
    timeout 120 $P 128 2 55 5 1 | tail -2 | head -1
  done
  timeout 120 $P 32 2 55 3 1 | grep -E 'slot 1|step (1[0-3]):|level 0'
  timeout 120 $P 160 2 55 3 1 | grep -E 'slot 1|step (1[0-3]):|level 0'
  timeout 300 $P 0 2 55 0 0 | tail -8
} > gpurun_out/r6d/conv3h_pf2.txt 2>&1
cat gpurun_out/r6d/conv3h_pf2.txt | cut -c1-220
