#!/bin/bash
# conv_down6_kernel: 96-channel workgroups where Cout % 96 == 0 (ASX_DOWN6_WIDE=1, default) against 48-channel ones everywhere; tests
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6h
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "down_conv or conv_layers" 2>&1 | grep -v "^$" | tail -14 | tee $O/pytest_down6.txt
run() {
  timeout 600 python bench.py --steps 3 --warmup 1 --no-arith-ab --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['kernel_ms'])"
}
(run wide; ASX_DOWN6_WIDE=0 run narrow; run wide; ASX_DOWN6_WIDE=0 run narrow) | tee $O/bench_down6_wide_ab.txt
