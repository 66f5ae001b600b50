#!/bin/bash
# conv_up6_kernel (transposed stride-2 conv on the 16-bit pipe, bf16 x 6): tests, bench A/B
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6h
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "up_conv or down_conv or conv_layers or net_ or excerpt" 2>&1 | grep -v "^$" | tail -22 | tee $O/pytest_up6.txt
run() {
  timeout 600 python bench.py --steps 3 --warmup 1 --no-arith-ab --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['kernel_ms'])"
}
(run up6_on; ASX_UP6=0 run up6_off; run up6_on; ASX_UP6=0 run up6_off) | tee $O/bench_up6_ab.txt
