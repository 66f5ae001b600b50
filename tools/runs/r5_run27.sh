#!/bin/bash
# level 1 (96 channels) on conv_wino6_kernel now that it runs fp16 x 3?  ASX_WINO6 = 144 (default) / 96, one call
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5C
mkdir -p $O
cd $GRAFT_REPO_ROOT
for w in 144 96 144 96; do
  ASX_WINO6=$w timeout 300 python bench.py --steps 6 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0 --traffic stored > $O/b_$w.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/b_$w.json')); pl=d['roofline']['per_level']['conv3x3']; print('ASX_WINO6=$w', d['value'], d['ms_per_step'], d['kernel_ms']['conv3x3'], {k:(v['kernel'][:17], v['avg_launch_ms']) for k,v in pl.items() if k in ('L1','L2')})" | tee -a $O/wino6_threshold.txt
done
