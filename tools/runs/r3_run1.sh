#!/bin/bash
# round-3 GPU run 1: halo-tile conv kernel -- parity (Demucs v4 / v3 / VR goldens + public layouts), A/B against gg_kernel
set -u
O=gpurun_out/r3a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_demucs.py tests/test_gpu_hdemucs.py tests/test_gpu_vr.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest_halo.log 2>&1; echo "rc=$?" >> $O/pytest_halo.log
tail -5 $O/pytest_halo.log
S="python tools/bench_siblings.py --workloads vr,htdemucs,hdemucs --cpu 0 --steps 2"
timeout 400 $S > $O/sib_halo.jsonl 2> $O/sib_halo.err
ASX_HALO=0 timeout 400 $S > $O/sib_gg.jsonl 2> $O/sib_gg.err
ASX_HALO_NT=128 timeout 400 $S > $O/sib_halo128.jsonl 2> $O/sib_halo128.err
ASX_HALO_NT=96 timeout 400 $S > $O/sib_halo96.jsonl 2> $O/sib_halo96.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r3a/sib_*.jsonl')):
    for l in open(f):
        try:
            r=json.loads(l); print(os.path.basename(f), r['config']['workload'][:20], r['value'], r['ms_per_step'], r['roofline']['kernel'][:30], r['roofline']['frac'], r.get('kernel_ms'))
        except Exception as e: print(f,'ERR',e, l[:100])
PY
ASX_PROF_DUMP=1 timeout 300 python tools/probe_demucs.py 60 8 2 > $O/dump_ht.log 2> $O/dump_ht.err
tail -12 $O/dump_ht.log
