#!/bin/bash
# balanced frame groups, stft3p at three workgroups per CU: parity + timing
set -u
O=gpurun_out/r2q
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_separate.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -5
for cfg in "0 0" "16 16" "0 8"; do
  set -- $cfg
  ASX_FFT3_G=$1 ASX_FFT3_GS=$2 timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 > $O/b_$1_$2.json 2>$O/b_$1_$2.err
  python - <<PY
import json
r=json.loads(open('$O/b_$1_$2.json').read().strip().splitlines()[-1])
k=r['kernel_ms']; print('G=$1 GS=$2', r['value'], {x:k[x] for x in k if 'stft' in x}, r['stage_roofline']['stft']['frac'], r['stage_roofline']['istft']['frac'])
PY
done
