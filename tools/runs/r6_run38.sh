#!/bin/bash
# EXPERIMENTAL build (python build.py --experimental): the tests that skip on the default library -- pair images, superseded Winograd generations
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6g
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_roformer.py -x -q -m gpu -k "pair_image or winograd or rowgemm or tdf" 2>&1 | tail -8 | tee $O/pytest_experimental.txt
