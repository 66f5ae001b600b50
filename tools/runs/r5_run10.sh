#!/bin/bash
# box-variance / flakiness record: the driver's three commands once more on a fresh box
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5j_$1
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu -W default 2>&1 | tail -6 | tee $O/pytest_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/pytest_tail.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_n1.json')); print(d['value'], d['ms_per_step'], d['kernel_ms'], {k:(v.get('value')) for k,v in d.get('siblings',{}).items()}, d['file_level']['rtf'])" | tee -a $O/pytest_tail.txt
