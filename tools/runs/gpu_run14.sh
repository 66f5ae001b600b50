#!/bin/bash
# packed-fp32 FFT arithmetic: parity + timing
set -u
O=gpurun_out/r2n
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_separate.py -q -x 2>&1 | tail -5
for cfg in "1 16" "1 8" "1 12" "0 16"; do
  set -- $cfg
  ASX_FFT3P=$1 ASX_FFT3_G=$2 timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 > $O/b_$1_$2.json 2>$O/b_$1_$2.err
  python - <<PY
import json
r=json.loads(open('$O/b_$1_$2.json').read().strip().splitlines()[-1])
k=r['kernel_ms']; print('P=$1 G=$2', r['value'], {x:k[x] for x in k if 'stft' in x or 'fin' in x or 'ola' in x}, r['stage_roofline']['stft'], r['stage_roofline']['istft'])
PY
done
