#!/bin/bash
# final: full GPU suite + smoke + default bench line + rocprof stats
set -u
O=gpurun_out/r2x
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 > $GRAFT_REPO_ROOT/$O/stats_bench.json 2> $GRAFT_REPO_ROOT/$O/stats.log
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r2x/bench_default.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['cpu_baseline']['value'], r.get('parity_rel_rms_vs_cpu'))
print({k:(v.get('value'), v.get('roofline',{}).get('frac')) for k,v in r['siblings'].items()})
print({k:v['frac'] for k,v in r['stage_roofline'].items()})
PY
