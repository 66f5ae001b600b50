#!/bin/bash
# round-3 GPU run 16: intensity threshold of the small-tile gather-GEMM dispatch, higher settings
set -u
O=gpurun_out/r3p
mkdir -p $O
S="python tools/bench_siblings.py --cpu 0 --steps 2"
for t in 200 1000000; do
  ASX_GG_LOWAI=$t timeout 600 $S --workloads htdemucs,hdemucs,vr > $O/sib_ai$t.jsonl 2> $O/sib_ai$t.err
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r3p/sib_*.jsonl')):
    for l in open(f):
        try:
            r=json.loads(l); print(os.path.basename(f), r['config']['workload'][:16], r['value'], r['ms_per_step'], {k[:20]:v for k,v in r['kernel_ms'].items() if v>10})
        except Exception as e: print(f,'ERR',e, l[:100])
PY
