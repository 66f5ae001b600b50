#!/bin/bash
# bf16x6 row GEMM integrated: parity suites that use linears, then the bench (siblings on) with the switch on and off
set -u
O=gpurun_out/r4k
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_roformer.py tests/test_gpu_demucs.py tests/test_gpu_hdemucs.py tests/test_gpu_mdxc.py -m gpu -q -x > $O/pytest_sel.log 2>&1; echo "rc=$?" >> $O/pytest_sel.log
tail -6 $O/pytest_sel.log
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --traffic stored > $O/bench_on.json 2> $O/bench_on.err
ASX_GEMM_BF16X6=0 timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --traffic stored --cpu-seconds 0 --file-level 0 > $O/bench_off.json 2> $O/bench_off.err
python - <<'PY'
import json
for f in ("on","off"):
    try:
        r=json.loads(open(f'gpurun_out/r4k/bench_{f}.json').read().strip().splitlines()[-1])
        print(f, r['value'], r['ms_per_step'], r.get('parity_rel_rms_vs_cpu'))
        print(r['kernel_ms'])
        print(r['stage_roofline'].get('tdf'))
        print({k:(v.get('value'), v.get('parity_rel_rms_vs_cpu')) for k,v in r.get('siblings',{}).items()})
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 $O/bench_on.err
