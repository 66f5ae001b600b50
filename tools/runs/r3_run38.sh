#!/bin/bash
# ring of two vs three buffers at the odd plane stride, same box
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
ASX_WINO_CFG=6 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "conv3x3_winograd and 3" 2>&1 | tail -1
for c in 5 6 5 6; do ASX_WINO_CFG=$c WINO=3 timeout 200 python tools/probe_wino.py 2>&1 | grep WINO | cut -c1-100 | sed "s/^/CFG=$c /"; done
