#!/bin/bash
# finer weight exponents (row GEMM: per four output columns; Winograd: per output channel): harness checks, then the whole GPU suite and the bench
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5E
mkdir -p $O
cd $GRAFT_REPO_ROOT
( timeout 200 tools/proto_gemm3 0 0 7 0 1 1 0 2>&1 | cut -c1-215
  timeout 200 tools/proto_gemm3 0 12 18 0 1 1 1 2>&1 | cut -c1-215
  timeout 200 tools/experimental/proto_wino6 0 0 13 0 256 1 1 2>&1 | cut -c1-250 ) | grep -v amdgpu.ids | tee $O/harness.txt
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_tail.txt
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --file-level 0 --traffic stored > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['kernel_ms'], {k:(v.get('value')) for k,v in d.get('siblings',{}).items()})" | tee -a $O/pytest_tail.txt
