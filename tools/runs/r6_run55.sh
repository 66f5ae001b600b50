#!/bin/bash
# conv3h_kernel on the deeper levels too? thresholds 144 (default) / 192 / 240 / 288: n x n launches over 48-channel slices (16 / 25 / 36 per layer) against conv_wino6_kernel
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6h
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() {
  timeout 600 python bench.py --steps 3 --warmup 1 --no-arith-ab --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
pl=d['roofline']['per_level']['conv3x3']
print('$1', d['value'], d['ms_per_step'], d['kernel_ms']['conv3x3'], {k: (v['avg_launch_ms'], v['kernel'][:14]) for k, v in pl.items()}, d.get('parity_rel_rms_vs_cpu'))"
}
(run thr144; ASX_CONV3H=192 run thr192; ASX_CONV3H=240 run thr240; ASX_CONV3H=288 run thr288; run thr144) | tee $O/bench_conv3h_deeper.txt
