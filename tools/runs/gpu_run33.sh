#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_gpu_vr.py tests/test_gpu_separate.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3
timeout 600 python tools/probe_vr.py 240 2>/dev/null | grep -E "audio|kernel ms"
python - <<'PY'
import sys, types, json
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import bench_siblings as BS
a = types.SimpleNamespace(seconds=240.0, steps=3, warmup=1, cpu=0)
r = BS.run_vr(a); print('vr', r['value'], r['ms_per_step'], r['roofline']['share_of_step_ms'])
PY
