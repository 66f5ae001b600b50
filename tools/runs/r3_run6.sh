#!/bin/bash
# round-3 GPU run 6: GLU row order with 16-byte epilogue accesses (Demucs); VR batch size sweep
set -u
O=gpurun_out/r3f
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_demucs.py tests/test_gpu_hdemucs.py tests/test_gpu_fullsize.py tests/test_gpu_separate.py tests/test_gpu_sharding.py -q -x -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
S="python tools/bench_siblings.py --cpu 0 --steps 2"
timeout 600 $S --workloads htdemucs,hdemucs > $O/sib_default.jsonl 2> $O/sib_default.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r3f/sib_*.jsonl')):
    for l in open(f):
        try:
            r=json.loads(l); print(os.path.basename(f), r['config']['workload'][:18], r['value'], r['ms_per_step'], r['roofline']['frac'], {k[:22]:v for k,v in (r.get('kernel_ms') or {}).items() if v>5})
        except Exception as e: print(f,'ERR',e, l[:100])
PY
for mb in 8 12 14 21; do
  ASX_HALO_MINBLK=600 timeout 300 python tools/probe_vr.py 240 $mb > $O/vr_mb$mb.log 2>&1; grep "audio\|kernel ms" $O/vr_mb$mb.log
done
