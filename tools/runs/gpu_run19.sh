#!/bin/bash
# round-2 evidence, final configuration: full GPU suite, rocprofv3 stats + PMC passes of the bench command, default bench line
set -u
O=gpurun_out/r2s
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 > $GRAFT_REPO_ROOT/$O/stats_bench.json 2> $GRAFT_REPO_ROOT/$O/stats.log
cd $GRAFT_REPO_ROOT
bash tools/pmc_run.sh $O/pmc bench.py --steps 1 --warmup 1 --cpu-seconds 0 --siblings 0
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r2s/bench_default.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['cpu_baseline'], r.get('parity_rel_rms_vs_cpu'))
print({k:(v.get('value'), v.get('roofline',{}).get('frac')) for k,v in r['siblings'].items()})
print(r['stage_roofline'])
PY
head -12 $O/pmc_summary.txt
