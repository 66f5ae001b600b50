#!/bin/bash
# round-2 GPU run 1: full GPU suite (default + TDF2 path), A/B of the row-GEMM variants on the bench workload, bench with siblings
set -u
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
python -c "import torch; print('devices', torch.cuda.device_count(), torch.cuda.get_device_name(0))" > $O/host.txt 2>&1
nproc >> $O/host.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
ASX_TDF2=3 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_roformer.py tests/test_gpu_demucs.py tests/test_gpu_mdxc.py tests/test_gpu_fullsize.py tests/test_gpu_vr.py -m gpu -x -q > $O/pytest_tdf2.log 2>&1; echo "rc=$?" >> $O/pytest_tdf2.log
for m in 0 1 2 3; do
  ASX_TDF2=$m timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 > $O/bench_tdf2_$m.json 2> $O/bench_tdf2_$m.err
done
ASX_TDF2=3 ASX_TDF2_SBIT=3 timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 > $O/bench_tdf2_3_sbit3.json 2> $O/bench_tdf2_3_sbit3.err
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 > $O/bench_forced_dist.json 2> $O/bench_forced_dist.err
timeout 120 python bench.py --gpus 2 --steps 1 --warmup 0 --cpu-seconds 0 --siblings 0 > $O/bench_gpus2.out 2>&1; echo "rc=$?" >> $O/bench_gpus2.out
tail -3 $O/pytest_gpu.log; tail -3 $O/pytest_tdf2.log
for f in $O/bench_tdf2_*.json; do python - "$f" <<'PY'
import json,sys
try:
    r=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], r['value'], r['ms_per_step'], {k:r['kernel_ms'][k] for k in ('tdf','down','up','conv3x3')}, r['stage_roofline']['tdf']['frac'], r['stage_roofline']['down']['frac'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
