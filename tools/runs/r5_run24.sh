#!/bin/bash
# attention on the fp16 x 3 arithmetic inside the engine: sibling lines with the option off / on in one call, then the whole GPU suite
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5x}
mkdir -p $O
cd $GRAFT_REPO_ROOT
for h in 0 1; do
  ASX_GEMM_F16X3=$h timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --file-level 0 --traffic stored > $O/bench_h$h.json 2> $O/bench_h$h.err; echo "bench h=$h rc=$?"
  python -c "
import json; d=json.load(open('$O/bench_h$h.json')); print('f16x3=$h', d['value'], d['ms_per_step'], {k:(v.get('value'), v.get('ms_per_step')) for k,v in d.get('siblings',{}).items()})" | tee -a $O/ab.txt
done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_tail.txt
