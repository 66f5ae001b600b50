#!/bin/bash
# round-3 GPU run 10: MDXC / VR device-resident file paths + concurrent container writes: tests, file-level rates
set -u
O=gpurun_out/r3j
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_separate.py tests/test_abi.py -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python tools/probe_file_level.py > $O/file_level_htdemucs.json 2> $O/file_level_htdemucs.err
cat $O/file_level_htdemucs.json; tail -3 $O/file_level_htdemucs.err
timeout 600 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --siblings 0 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r3j/bench.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step']); print(json.dumps(r.get('file_level'))[:1500])
PY
