#!/bin/bash
# the level-0 first TDF linear (N = 384) on ONE 8-wave workgroup per row block (ASX_TDF3_NW8=1, default) against two 4-wave workgroups; tests first
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6h
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tdf or rowgemm or net_ or hq3_excerpt_vs_oracle" 2>&1 | grep -v "^$" | tail -3 | tee $O/pytest_nw8.txt
run() {
  timeout 600 python bench.py --steps 3 --warmup 1 --no-arith-ab --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['kernel_ms']['tdf'], d['roofline']['per_level']['tdf']['L0.F_to_F8']['avg_launch_ms'])"
}
(run nw8; ASX_TDF3_NW8=0 run nw4; run nw8; ASX_TDF3_NW8=0 run nw4) | tee $O/bench_nw8_ab.txt
