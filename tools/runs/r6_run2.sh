#!/bin/bash
# round 6, run 2: conv3h_kernel with two tiles of loads in flight, the incremental band walk and the read-ahead consumer schedule
mkdir -p gpurun_out/r6a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=tools/proto_conv3h
{
  timeout 300 $P 0 2 55 5
  for abl in 2 4 8 6 14; do timeout 120 $P $abl 2 55 5 | tail -2; done
} > gpurun_out/r6a/conv3h_run2.txt 2>&1
tail -40 gpurun_out/r6a/conv3h_run2.txt
