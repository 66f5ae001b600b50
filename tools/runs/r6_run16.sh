#!/bin/bash
# round 6, run 16: conv3h_kernel inside the engine -- its op tests, the HQ_3 excerpt in every kernel mode, the whole-song digests, bench line
mkdir -p gpurun_out/r6b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv3x3_direct or winograd_hq3_excerpt or test_conv_layers" > gpurun_out/r6b/pytest_conv3h.txt 2>&1
tail -5 gpurun_out/r6b/pytest_conv3h.txt
timeout 900 python -m pytest tests/test_gpu_fullsong.py -x -q -k "mdx" > gpurun_out/r6b/pytest_fullsong.txt 2>&1
tail -5 gpurun_out/r6b/pytest_fullsong.txt
timeout 600 python bench.py > gpurun_out/r6b/bench_conv3h.json 2> gpurun_out/r6b/bench_conv3h.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6b/bench_conv3h.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], json.dumps(d.get("roofline", {}).get("per_level", {}).get("conv3x3", {}))[:1500])
PY
ASX_CONV3H=0 timeout 600 python bench.py --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('conv3h off:', d['value'], d['ms_per_step'])"
