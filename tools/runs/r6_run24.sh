#!/bin/bash
# round 6, run 24: driver-style sequence on the final tree: GPU suite, smoke(), bench line
mkdir -p gpurun_out/r6e; rm -f gpurun_out/r6e/*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6e/pytest_gpu.txt 2>&1
tail -3 gpurun_out/r6e/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6e/smoke.txt 2>&1; tail -4 gpurun_out/r6e/smoke.txt
timeout 900 python bench.py > gpurun_out/r6e/bench_n1.json 2> gpurun_out/r6e/bench_n1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6e/bench_n1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("arithmetic_ab", {}).get("value"), d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["traffic"])
print(d["kernel_ms"], d.get("parity_rel_rms_vs_cpu"))
print({k: v["avg_launch_ms"] for k, v in d["roofline"]["per_level"]["conv3x3"].items()})
PY
