#!/bin/bash
# round 6, run 17: whole GPU suite + driver-style bench line + kernel stats on the tree with conv3h_kernel in the engine
mkdir -p gpurun_out/r6b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6b/pytest_gpu.txt 2>&1
tail -6 gpurun_out/r6b/pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/r6b/bench_n1.json 2> gpurun_out/r6b/bench_n1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6b/bench_n1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["dtype"][:60], d.get("arithmetic_ab"))
print(json.dumps(d["roofline"]["per_level"]["conv3x3"]["L0"]))
print({k: (v["value"], v["roofline"].get("frac")) for k, v in d.get("siblings", {}).items() if isinstance(v, dict) and "value" in v})
PY
