#!/bin/bash
# running exponents that rise again (tdf3_kernel<H> rows, conv_wino6_kernel<H> tile rows): stress tests, timing of the row GEMM shapes, then the whole GPU suite + bench
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6g
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "block_exponent or rowgemm or winograd_bf16x6" 2>&1 | grep -v "^$" | tail -25 | tee $O/pytest_exponent.txt
for abl in 0 0; do
  timeout 300 tools/proto_gemm3 $abl 4 10 0 1 0 0 2>&1 | grep -v "amdgpu.ids" | awk '{print $1,$2,$3,$4,$5,$6,$12,$13,$14,$15,$16,$17}'
done | tee $O/tdf3h_rise_timing.txt
timeout 300 tools/proto_gemm3 0 16 18 0 1 0 0 2>&1 | grep -v "amdgpu.ids" | tee -a $O/tdf3h_rise_timing.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6g/bench_n1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("arithmetic_ab", {}).get("value"), d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["traffic"])
print(d["kernel_ms"], d.get("parity_rel_rms_vs_cpu"))
print({k: v["avg_launch_ms"] for k, v in d["roofline"]["per_level"]["conv3x3"].items()})
PY
