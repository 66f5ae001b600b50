#!/bin/bash
# round 6, run 5: vector-memory front-end micro-benchmark (cycles per wave-instruction by load width and lane pattern)
mkdir -p gpurun_out/r6a
cd $GRAFT_REPO_ROOT
timeout 120 tools/experimental/micro_ta > gpurun_out/r6a/micro_ta.txt 2>&1
cat gpurun_out/r6a/micro_ta.txt
