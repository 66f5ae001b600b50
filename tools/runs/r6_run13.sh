#!/bin/bash
mkdir -p gpurun_out/r6a
cd $GRAFT_REPO_ROOT
timeout 200 tools/experimental/micro_fetch > gpurun_out/r6a/micro_fetch.txt 2>&1
cat gpurun_out/r6a/micro_fetch.txt
