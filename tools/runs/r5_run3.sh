#!/bin/bash
# which change broke the bf16 x 6 leg of the full-depth BS-Roformer chunk (6.97e-5 instead of 1.5e-6)?  A/B of the tile map on one box,
# then the rest of the suite without -x
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5c
mkdir -p $O
cd $GRAFT_REPO_ROOT
for m in 0 11 10 default; do
  if [ $m = default ]; then unset ASX_TDF3_MAP; else export ASX_TDF3_MAP=$m; fi
  timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "bs_roformer" 2>&1 | grep "BS-Roformer\|passed\|failed" | sed "s/^/map=$m: /" | tee -a $O/ab.log
done
unset ASX_TDF3_MAP
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -n "whole workload\|passed\|failed\|FAILED" $O/pytest.log | tail -40
