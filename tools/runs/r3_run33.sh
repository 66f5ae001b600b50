#!/bin/bash
# Winograd stage / ring configurations (ASX_WINO_CFG): correctness of each + per-launch times
set -u
O=gpurun_out/r3w
export PYTHONPATH=$GRAFT_REPO_ROOT
for c in 1 2 3; do
  ASX_WINO_CFG=$c timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "conv3x3_winograd and 3" 2>&1 | tail -1 | sed "s/^/CFG=$c /"
done
for c in 0 1 2 3; do ASX_WINO_CFG=$c WINO=3 timeout 200 python tools/probe_wino.py 2>&1 | grep WINO | cut -c1-330 | sed "s/^/CFG=$c /" | tee -a $O/launches6.log; done
