#!/bin/bash
# round-3 GPU run 2: device-resident file path (tests + bench file_level), VR per-launch dump
set -u
O=gpurun_out/r3b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_separate.py tests/test_abi.py -q -x > $O/pytest_sep.log 2>&1; echo "rc=$?" >> $O/pytest_sep.log
tail -15 $O/pytest_sep.log
timeout 600 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --siblings 0 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r3b/bench.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step']); print(json.dumps(r.get('file_level'), indent=1))
PY
tail -5 $O/bench.err
ASX_PROF_DUMP=1 timeout 300 python tools/probe_vr.py 60 8 > $O/dump_vr.log 2> $O/dump_vr.err
tail -12 $O/dump_vr.log
