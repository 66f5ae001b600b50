#!/bin/bash
# fp16 x 3 row GEMM inside the engine: its parity tests, the bench song with the option off / on in one call (box variance), the sibling
# lines both ways, then the whole GPU suite on the new default
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5n
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rowgemm or tdf" -s 2>&1 | grep -v "^$" | tail -30 | tee $O/pytest_rowgemm.txt
for h in 0 1; do
  ASX_GEMM_F16X3=$h timeout 600 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --file-level 0 --traffic stored > $O/bench_h$h.json 2> $O/bench_h$h.err; echo "bench h=$h rc=$?"
  python -c "
import json; d=json.load(open('$O/bench_h$h.json')); print('f16x3=$h', d['value'], d['ms_per_step'], d['kernel_ms'], {k:(v.get('value'), v.get('ms_per_step')) for k,v in d.get('siblings',{}).items()}, d.get('parity_rel_rms_vs_cpu'))" | tee -a $O/ab.txt
done
timeout 1200 python -m pytest tests -x -q -m gpu -W default 2>&1 | tail -8 | tee $O/pytest_tail.txt
