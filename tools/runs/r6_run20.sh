#!/bin/bash
# round 6, runs 20 / 22: evidence on the tree with conv3h_kernel on levels 0-2 (22: default build without the superseded kernel generations): GPU suite, driver-style bench line, kernel stats, PMC record of conv3h_kernel
mkdir -p gpurun_out/r6d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6d/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r6d/pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/r6d/bench_n1.json 2> gpurun_out/r6d/bench_n1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6d/bench_n1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("arithmetic_ab", {}).get("value"), d["roofline"]["kernel"][:40], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"][:60])
print(d["kernel_ms"])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6d/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored --no-arith-ab > /dev/null 2>&1)
find gpurun_out/r6d/stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/r6d/kernel_stats_bench.csv
head -12 gpurun_out/r6d/kernel_stats_bench.csv | cut -c1-200
bash tools/pmc_bin.sh gpurun_out/r6d/pmc_conv3h tools/proto_conv3h 0 2 55 3 1
python3 tools/pmc_kernel_json.py gpurun_out/r6d/pmc_conv3h conv3h $((55*48*256*3072*4*2)) "tools/pmc_bin.sh on tools/proto_conv3h 0 2 55 3 1 (level-0 shape, 55 chunks; rocprofv3 --pmc passes, one counter group per run)" > gpurun_out/r6d/pmc_conv3h.json
cat gpurun_out/r6d/pmc_conv3h.json | head -30
rm -rf gpurun_out/r6d/stats
