#!/bin/bash
# sinc resampler hook (all ratios, mono / stereo), MDXC pitch_shift, VR + MDXC + sharding suites
set -u
O=gpurun_out/r4f
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_vr.py tests/test_gpu_mdxc.py tests/test_gpu_sharding.py tests/test_gpu_separate.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
