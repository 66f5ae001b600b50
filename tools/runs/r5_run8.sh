#!/bin/bash
# Roformer STFT options at the engine level, the VR chain with one-channel sinc calls (librosa >= 0.10), whole-workload digests
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5h
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_roformer.py tests/test_gpu_vr.py tests/test_gpu_fullsong.py tests/test_gpu_fullsize.py -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -n "whole workload .*worst\|BS-Roformer ep\|VR 4band\|error energy\|passed\|failed\|FAILED\|^E " $O/pytest.log | tail -30
