#!/bin/bash
set -u
O=gpurun_out/r4g
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_separate.py tests/test_gpu_sharding.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
