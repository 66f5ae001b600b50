#!/bin/bash
# round-3 GPU run 14: 128-row narrow-N gather-GEMM tiles (A/B)
set -u
O=gpurun_out/r3n
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_demucs.py tests/test_gpu_hdemucs.py tests/test_gpu_vr.py -q -x -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
S="python tools/bench_siblings.py --cpu 0 --steps 2"
timeout 600 $S --workloads htdemucs,hdemucs,vr > $O/sib_m128.jsonl 2> $O/sib_m128.err
ASX_GG_M128=0 timeout 600 $S --workloads htdemucs,hdemucs,vr > $O/sib_m256.jsonl 2> $O/sib_m256.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r3n/sib_*.jsonl')):
    for l in open(f):
        try:
            r=json.loads(l); print(os.path.basename(f), r['config']['workload'][:16], r['value'], r['ms_per_step'], r['roofline'].get('hbm_bound_launches'), {k[:20]:v for k,v in r['kernel_ms'].items() if v>10})
        except Exception as e: print(f,'ERR',e, l[:100])
PY
