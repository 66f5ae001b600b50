#!/bin/bash
# round 4, first contact: whole GPU suite (ABI 5: VR sinc converter, whole-song digest test), the new smoke, default bench line
set -u
O=gpurun_out/r4a
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r4a/bench_n1.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['algorithmic'], r['roofline']['direct_kernel']['frac'], r.get('parity_rel_rms_vs_cpu'))
print(r['kernel_ms'])
print({k:v.get('value') for k,v in r['siblings'].items()}, r['file_level'].get('rtf'))
PY
