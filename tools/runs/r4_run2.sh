#!/bin/bash
# weight-stationary Winograd kernel: parity, then A/B inside one call (ASX_WINOS=0 / 1) on the bench workload
set -u
O=gpurun_out/r4b
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "winograd or batching or conv" > $O/pytest_wino.log 2>&1; echo "rc=$?" >> $O/pytest_wino.log
tail -8 $O/pytest_wino.log
for w in 1 0 1; do
  ASX_WINOS=$w timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --cpu-seconds 0 --siblings 0 --file-level 0 > $O/bench_winos$w.json 2> $O/bench_winos$w.err
  python - <<PY
import json
r=json.loads(open('$O/bench_winos$w.json').read().strip().splitlines()[-1])
print('WINOS=$w', r['value'], r['ms_per_step'], r['roofline']['frac'], r['kernel_ms'])
PY
done
ASX_PROF_DUMP=1 timeout 600 python bench.py --gpus 1 --steps 1 --warmup 1 --cpu-seconds 0 --siblings 0 --file-level 0 2> $O/prof_dump.err > /dev/null
grep -i "conv3x3\|cls 0\|cls=0" $O/prof_dump.err | head -80 > $O/prof_dump_conv.txt; wc -l $O/prof_dump_conv.txt
