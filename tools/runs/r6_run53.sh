#!/bin/bash
# conv_up6_kernel: six virtual tiles per workgroup (36 KB of LDS, four workgroups per CU; default) against four (30 KB, five workgroups; the input re-read by 1.5x as many groups)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6h
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() {
  timeout 600 python bench.py --steps 3 --warmup 1 --no-arith-ab --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['kernel_ms']['up'])"
}
(run nrep6; ASX_UP_NREP=4 run nrep4; run nrep6; ASX_UP_NREP=4 run nrep4) | tee $O/bench_up6_nrep_ab.txt
