#!/bin/bash
# default bench line with the live PMC traffic measurement (driver's command)
set -u
O=gpurun_out/r4h
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
SECONDS=0; timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r4h/bench_default.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic'], r['roofline']['algorithmic_bytes_per_launch'])
print(r['roofline']['traffic_source'])
print({k:v.get('value') for k,v in r['siblings'].items()}, r['file_level'].get('rtf'))
PY
echo "bench wall: $SECONDS s"
