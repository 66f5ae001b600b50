#!/bin/bash
# round-3 GPU run 24: radix-16 / radix-8 passes in the generic Stockham FFT (every geometry but MDX 6144 / 1024): whole GPU suite, A/B
set -u
O=gpurun_out/r3x
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
S="python tools/bench_siblings.py --cpu 0 --steps 2 --workloads htdemucs,hdemucs,roformer,vr"
timeout 900 $S > $O/sib_r16.jsonl 2> $O/sib_r16.err
ASX_FFT_RADIX4=1 timeout 900 $S > $O/sib_r4.jsonl 2> $O/sib_r4.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r3x/sib_*.jsonl')):
    for l in open(f):
        r=json.loads(l); print(os.path.basename(f), r['config']['workload'][:14], r['value'], r['ms_per_step'], {k[:10]:v for k,v in r['kernel_ms'].items() if k in ('istft','ola','stft')})
PY
