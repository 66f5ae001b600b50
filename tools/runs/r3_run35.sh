#!/bin/bash
# odd plane stride (ASX_WINO_CFG=4): correctness + A/B on one box
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
ASX_WINO_CFG=4 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "conv3x3_winograd and 3" 2>&1 | tail -3 | cut -c1-300
for c in 0 4 0 4; do ASX_WINO_CFG=$c WINO=3 timeout 200 python tools/probe_wino.py 2>&1 | grep WINO | cut -c1-120 | sed "s/^/CFG=$c /"; done
