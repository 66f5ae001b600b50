#!/bin/bash
# Winograd v2 (in-register input transform): parity tests + per-class time against the direct kernel (240-s song)
mkdir -p gpurun_out/r3w
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k winograd > gpurun_out/r3w/tests.log 2>&1
tail -3 gpurun_out/r3w/tests.log
for w in 0 2 3 1; do
  WINO=$w timeout 300 python tools/probe_perf.py 240 64 2>&1 | grep -E "^audio|^conv3x3" | sed "s/^/WINO=$w /" | tee -a gpurun_out/r3w/perf.log
done
