#!/bin/bash
# round-3 GPU run 5: halo weight image without ds_read2_b64 conflicts, XCD-aware attention grids
set -u
O=gpurun_out/r3e
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_demucs.py tests/test_gpu_hdemucs.py tests/test_gpu_roformer.py tests/test_gpu_vr.py tests/test_gpu_fullsize.py -q -x -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
S="python tools/bench_siblings.py --cpu 0 --steps 2"
timeout 600 $S --workloads htdemucs,hdemucs,roformer,vr > $O/sib_default.jsonl 2> $O/sib_default.err
ASX_HALO_MINBLK=600 timeout 600 $S --workloads vr,hdemucs > $O/sib_minblk600.jsonl 2> $O/sib_minblk600.err
ASX_HALO_MINBLK=300 timeout 600 $S --workloads vr > $O/sib_minblk300.jsonl 2> $O/sib_minblk300.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r3e/sib_*.jsonl')):
    for l in open(f):
        try:
            r=json.loads(l); print(os.path.basename(f), r['config']['workload'][:18], r['value'], r['ms_per_step'], r['roofline']['frac'], {k[:22]:v for k,v in (r.get('kernel_ms') or {}).items() if v>5})
        except Exception as e: print(f,'ERR',e, l[:100])
PY
bash tools/pmc_run.sh $O/pmc_ht tools/probe_demucs.py 30 8 2
python tools/pmc_summary.py $O/pmc_ht > $O/pmc_ht_summary.txt 2>&1
grep -A10 "hg_kernel<6, 3>\|mha_kernel" $O/pmc_ht_summary.txt | head -40
