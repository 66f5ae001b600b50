#!/bin/bash
# whole GPU suite on the split-source tree (Roformer STFT options, captured graph with other engines coming and going, ConvTDFNet variants)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5g
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -n "whole workload .*worst\|BS-Roformer ep\|error energy\|passed\|failed\|FAILED\|^E " $O/pytest.log | tail -40
