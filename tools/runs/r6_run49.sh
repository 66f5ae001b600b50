#!/bin/bash
# conv_down6_kernel: tile height 4 from which channel count up?  ASX_DOWN6_TH4 = 1000 (never) / 48 (levels 1+, default) / 0 (everywhere); tests first
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6h
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "down_conv or conv_layers or net_" 2>&1 | grep -v "^$" | tail -3 | tee $O/pytest_down6_th.txt
ASX_DOWN6_TH4=0 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "down_conv or conv_layers" 2>&1 | grep -v "^$" | tail -2 | tee -a $O/pytest_down6_th.txt
run() {
  timeout 600 python bench.py --steps 3 --warmup 1 --no-arith-ab --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['kernel_ms']['down'], d['kernel_ms']['up'])"
}
(ASX_DOWN6_TH4=1000 run th2_everywhere; run th4_from_level1; ASX_DOWN6_TH4=0 run th4_everywhere; ASX_DOWN6_TH4=1000 run th2_everywhere; run th4_from_level1; ASX_DOWN6_TH4=0 run th4_everywhere) | tee $O/bench_down6_th_ab.txt
