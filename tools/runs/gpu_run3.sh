#!/bin/bash
# round-2 GPU run 3: fast FFT path (correctness + A/B), row-coalesced TDF epilogue A/B
set -u
O=gpurun_out/r2c
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log
tail -5 $O/pytest_parity.log
B="python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --siblings 0"
ASX_FFT3=0 timeout 300 $B > $O/b_fft3_off.json 2> $O/b_fft3_off.err
ASX_FFT3=1 timeout 300 $B > $O/b_fft3_on.json 2> $O/b_fft3_on.err
ASX_TDF2=1 ASX_TDF2_ABL=8 timeout 300 $B > $O/b_epi8_m1.json 2> $O/b_epi8_m1.err
ASX_TDF2=2 ASX_TDF2_ABL=8 timeout 300 $B > $O/b_epi8_m2.json 2> $O/b_epi8_m2.err
ASX_TDF2=2 timeout 300 $B > $O/b_m2.json 2> $O/b_m2.err
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
ASX_TDF2=2 ASX_TDF2_ABL=8 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_roformer.py tests/test_gpu_demucs.py tests/test_gpu_fullsize.py -m gpu -q > $O/pytest_epi8.log 2>&1; echo "rc=$?" >> $O/pytest_epi8.log
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r2c/b_*.json')):
    try:
        r=json.load(open(f)); km=r['kernel_ms']; print(os.path.basename(f), r['value'], r['ms_per_step'], {k:km.get(k) for k in ('stft','istft','ola','finalize','tdf','down')}, {k:r['stage_roofline'][k]['frac'] for k in ('stft','istft','tdf') if k in r['stage_roofline']})
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-500:])
PY
tail -4 $O/pytest_gpu.log; tail -4 $O/pytest_epi8.log
