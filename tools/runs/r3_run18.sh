#!/bin/bash
# round-3 GPU run 18: transformer linears on 64-row tiles (default), attention with four workgroups per CU (single-buffered) vs three (double-buffered)
set -u
O=gpurun_out/r3r
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_demucs.py tests/test_gpu_hdemucs.py -q -x -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
S="python tools/bench_siblings.py --cpu 0 --steps 2"
timeout 600 $S --workloads htdemucs,hdemucs > $O/sib_default.jsonl 2> $O/sib_default.err
ASX_MHA_DB=0 timeout 600 $S --workloads htdemucs,hdemucs > $O/sib_mha4wg.jsonl 2> $O/sib_mha4wg.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r3r/sib_*.jsonl')):
    for l in open(f):
        try:
            r=json.loads(l); print(os.path.basename(f), r['config']['workload'][:16], r['value'], r['ms_per_step'], {k[:20]:v for k,v in r['kernel_ms'].items() if v>10})
        except Exception as e: print(f,'ERR',e, l[:100])
PY
