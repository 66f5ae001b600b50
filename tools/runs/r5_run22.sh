#!/bin/bash
# profile record of the fp16 x 3 default: whole GPU suite, the driver's bench command, rocprofv3 kernel stats of the bench,
# PMC passes (conv_wino3 / conv_wino6 / tdf3 per layer)
set -u
export RUN_TAG=${1:-r5v}
O=$GRAFT_REPO_ROOT/gpurun_out/${RUN_TAG:-r5v}
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_tail.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python - <<'PY' | tee -a $O/pytest_tail.txt
import json, os
d = json.load(open("gpurun_out/" + os.environ.get("RUN_TAG", "r5v") + "/bench_n1.json"))
print(d["value"], d["ms_per_step"], d["kernel_ms"])
print({k: (v.get("value"), v.get("ms_per_step")) for k, v in d.get("siblings", {}).items()}, d.get("file_level", {}).get("rtf"))
print(d["stage_roofline"]["tdf"])
PY
(cd /tmp && export TMPDIR=/tmp && PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0 --traffic stored > $O/stats.log 2>&1)
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bench.csv; head -12 $O/kernel_stats_bench.csv | cut -c1-170
rm -rf $O/stats
bash tools/pmc_run.sh gpurun_out/${RUN_TAG:-r5v}/pmc_bench bench.py --pmc-child --seconds 240
python tools/pmc_summary.py $O/pmc_bench > $O/pmc_summary.txt 2>&1
HOW="rocprofv3 --pmc passes (tools/pmc_run.sh) of bench.py --pmc-child --seconds 240 (one warm-up + one demix of the bench song), round-5 tree with gemm_f16x3 = 1"
python tools/pmc_tdf3_json.py $O/pmc_bench --how "$HOW" > $O/pmc_tdf3h.json
python -c "
import json; d=json.load(open('$O/pmc_tdf3h.json')); print(d['arithmetic'])
for k,v in d['layers'].items(): print(k, v.get('tile'), v['dispatches'], v.get('traffic_over_algorithmic'), v.get('mfma_util'), v.get('lds_conflict_over_active'))
print(d.get('traffic_over_algorithmic_all_matched'))"
rm -rf $O/pmc_bench/*/p_agent_info.csv
du -sh $O
