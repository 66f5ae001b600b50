#!/bin/bash
# round 6, run 19: conv3h_kernel on levels 0 and 1 (96 channels as four launches): op tests, excerpt modes, whole-song digests, bench with 48 / 96 / 144 thresholds
mkdir -p gpurun_out/r6b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv3x3_direct or winograd_hq3_excerpt" > gpurun_out/r6b/pytest_conv3h_l1.txt 2>&1
tail -5 gpurun_out/r6b/pytest_conv3h_l1.txt
timeout 900 python -m pytest tests/test_gpu_fullsong.py -x -q -k "mdx" > gpurun_out/r6b/pytest_fullsong_l1.txt 2>&1
tail -3 gpurun_out/r6b/pytest_fullsong_l1.txt
for thr in 96 48 144; do
ASX_CONV3H=$thr timeout 600 python bench.py --steps 10 --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored --no-arith-ab 2>/dev/null > gpurun_out/r6b/bench_thr$thr.json
python - <<PY
import json
d = json.loads(open("gpurun_out/r6b/bench_thr$thr.json").read().strip().splitlines()[-1])
print("threshold $thr:", d["value"], d["ms_per_step"], {k: (v["avg_launch_ms"], v["kernel"][:14]) for k, v in d["roofline"]["per_level"]["conv3x3"].items()})
print("   roofline:", d["roofline"]["kernel"][:50], d["roofline"]["achieved"], d["roofline"]["frac"])
PY
done
