#!/bin/bash
# conv_down6_kernel / conv_up6_kernel, one stage buffer and four workgroups per CU: tests, bench with both on / both off
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6h
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "up_conv or down_conv or conv_layers or net_" 2>&1 | grep -v "^$" | tail -4 | tee $O/pytest_updown6.txt
run() {
  timeout 600 python bench.py --steps 3 --warmup 1 --no-arith-ab --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['kernel_ms'])"
}
(run on; ASX_UP6=0 ASX_DOWN6=0 run off; run on; ASX_UP6=0 ASX_DOWN6=0 run off) | tee $O/bench_updown6_ab.txt
