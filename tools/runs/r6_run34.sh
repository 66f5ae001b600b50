#!/bin/bash
# pair images wired through the engine: new tests, the row-GEMM / net / Roformer suites, bench A/B
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_roformer.py tests/test_gpu_mdxc.py -x -q -m gpu -k "tdf or rowgemm or pair or net_ or roformer or forward or demix or excerpt" 2>&1 | tail -15 | tee $O/pytest_pair.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-arith-ab 2>/dev/null | tail -1 > $O/bench_pair_on.json
ASX_PAIR_IMAGES=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-arith-ab 2>/dev/null | tail -1 > $O/bench_pair_off.json
python - <<'PY'
import json
for n in ("on","off"):
    d=json.load(open(f"gpurun_out/r6f/bench_pair_{n}.json"))
    print(n, d["value"], d["ms_per_step"], d["kernel_ms"])
PY
