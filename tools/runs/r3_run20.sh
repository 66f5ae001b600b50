#!/bin/bash
# round-3 GPU run 20: BS-Roformer feed-forward GELU on the branch-free erf (A/B), parity of that build
set -u
O=gpurun_out/r3t
mkdir -p $O
S="python tools/bench_siblings.py --cpu 0 --steps 2"
timeout 600 $S --workloads roformer > $O/sib_gelu2.jsonl 2> $O/sib_gelu2.err
ASX_ROF_GELU=6 timeout 600 $S --workloads roformer > $O/sib_gelu6.jsonl 2> $O/sib_gelu6.err
ASX_ROF_GELU=6 timeout 900 python -m pytest tests/test_gpu_roformer.py "tests/test_gpu_fullsize.py" -q -x -m gpu -k "rof or Rof or roformer" > $O/pytest_gelu6.log 2>&1; echo "rc=$?" >> $O/pytest_gelu6.log
tail -3 $O/pytest_gelu6.log
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r3t/sib_*.jsonl')):
    for l in open(f):
        r=json.loads(l); print(os.path.basename(f), r['value'], r['ms_per_step'], r['kernel_ms'])
PY
