#!/bin/bash
# round-3 GPU run 3: new exports (bag combine, finalize4, sync-free sibling calls), whole-song parity records, htdemucs_ft bag
set -u
O=gpurun_out/r3c
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_separate.py tests/test_gpu_demucs.py tests/test_gpu_hdemucs.py tests/test_gpu_roformer.py tests/test_gpu_parity.py tests/test_gpu_sharding.py tests/test_abi.py -q -x -m "gpu or not gpu" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 900 python tools/fullsong_parity.py > $O/fullsong_parity.json 2> $O/fullsong_parity.err
tail -6 $O/fullsong_parity.err | cut -c1-700
timeout 600 python tools/bench_siblings.py --workloads htdemucs_ft --cpu 0 --steps 2 > $O/sib_ft.jsonl 2> $O/sib_ft.err
cat $O/sib_ft.jsonl; tail -3 $O/sib_ft.err
timeout 600 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0 > $O/bench.json 2> $O/bench.err
ASX_FINALIZE4=0 timeout 600 python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0 > $O/bench_fin1.json 2> $O/bench_fin1.err
python - <<'PY'
import json
for f in ('bench','bench_fin1'):
    try:
        r=json.loads(open(f'gpurun_out/r3c/{f}.json').read().strip().splitlines()[-1])
        print(f, r['value'], r['ms_per_step'], r['kernel_ms'].get('finalize'), r['stage_roofline'].get('finalize'), r['rccl'])
    except Exception as e: print(f,'ERR',e)
PY
