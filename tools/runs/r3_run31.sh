#!/bin/bash
# Winograd as the default 3x3 kernel: whole GPU suite, per-launch times, whole-song parity of the MDX cases, short bench
set -u
O=gpurun_out/r3w
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log | cut -c1-300
WINO=3 timeout 200 python tools/probe_wino.py 2>&1 | grep WINO | cut -c1-200 | tee -a $O/launches5.log
timeout 900 python tools/fullsong_parity.py --cases mdx_hq3,mdx23c > $O/fullsong_parity_mdx.json 2> $O/fullsong_parity_mdx.err
tail -3 $O/fullsong_parity_mdx.err | cut -c1-400
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3w/fullsong_parity_mdx.json'))
for k,v in d.get('cases',{}).items():
    print(k, {s:(float('%.3g'%x['rel_rms']), float('%.3g'%x['abs_rms_err'])) for s,x in v['stems'].items()})
PY
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 12 --siblings 0 --file-level 1 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r3w/bench.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], json.dumps(r['roofline'])[:1500]); print(r.get('parity_rel_rms_vs_cpu'), r['file_level'].get('rtf'), r['kernel_ms'])
PY
tail -3 $O/bench.err
