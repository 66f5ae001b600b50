#!/bin/bash
# the GPU suite with every fresh device allocation filled with NaN bytes (ASX_POISON=255): nothing the new kernels read may be memory the engine never wrote
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6h
mkdir -p $O
cd $GRAFT_REPO_ROOT
ASX_POISON=255 timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_poison.txt 2>&1
tail -4 $O/pytest_gpu_poison.txt
