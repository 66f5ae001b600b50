#!/bin/bash
# persistent Winograd kernel (DMA ring across tile boundaries): parity + A/B on one box
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
ASX_WINO_PERSIST=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "conv3x3_winograd and 3" 2>&1 | tail -3 | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3 | cut -c1-300
for p in 2048 0 2048 0; do ASX_WINO_PERSIST=$p WINO=3 timeout 200 python tools/probe_wino.py 2>&1 | grep WINO | cut -c1-330 | sed "s/^/PERSIST=$p /" | tee -a gpurun_out/r3p/persist.log; done
