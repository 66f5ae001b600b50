#!/bin/bash
# EXPERIMENTAL build (python build.py --experimental): the whole GPU suite -- nothing skips (pair images, superseded Winograd generations, ablation switches)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6h
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_experimental.txt 2>&1
tail -4 $O/pytest_experimental.txt
