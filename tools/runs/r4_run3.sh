#!/bin/bash
# weight-stationary Winograd kernel: parity, then ablation builds on the bench workload (conv3x3 class ms per song)
set -u
O=gpurun_out/r4c
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "winograd or batching or conv" > $O/pytest_wino.log 2>&1; echo "rc=$?" >> $O/pytest_wino.log
tail -5 $O/pytest_wino.log
for abl in 0 1 2 4 5 13; do
  ASX_WINOS_ABL=$abl timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --cpu-seconds 0 --siblings 0 --file-level 0 > $O/bench_abl$abl.json 2> $O/bench_abl$abl.err
  python - <<PY
import json
r=json.loads(open('$O/bench_abl$abl.json').read().strip().splitlines()[-1])
print('ABL=$abl', r['ms_per_step'], r['kernel_ms']['conv3x3'])
PY
done
