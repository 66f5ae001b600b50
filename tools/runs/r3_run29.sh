#!/bin/bash
# Winograd third generation (4-buffer ring, b128 weight fragments): parity + per-launch times
mkdir -p gpurun_out/r3w
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "winograd and 3" > gpurun_out/r3w/tests3.log 2>&1
tail -3 gpurun_out/r3w/tests3.log
for w in 2 3; do WINO=$w timeout 200 python tools/probe_wino.py 2>&1 | grep WINO | tee -a gpurun_out/r3w/launches3.log; done
