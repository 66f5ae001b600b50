#!/bin/bash
# Round 5, second GPU call: the whole GPU suite on the wino6 tree (whole-workload digests of every BASELINE config included), the
# driver's bench command, rocprofv3 kernel stats of the bench, PMC passes (conv_wino3 / conv_wino6 / tdf3 per layer).
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -n "whole workload\|passed\|failed\|rror" $O/pytest.log | tail -40
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5b/bench_n1.json"))
print(d["value"], d["ms_per_step"], d["kernel_ms"])
r = d["roofline"]; print({k: r[k] for k in ("kernel", "achieved", "frac", "traffic", "share_of_step_ms", "conv3x3_class_ms")})
for k, v in r["per_level"]["conv3x3"].items(): print(k, v["kernel"][:18], v["avg_launch_ms"], v["frac"])
print({k: (v.get("value"), v.get("ms_per_step")) for k, v in d.get("siblings", {}).items()})
print(d.get("file_level", {}).get("rtf"), d.get("parity_rel_rms_vs_cpu"), d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
PY
(cd /tmp && export TMPDIR=/tmp && PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0 --traffic stored > $O/stats.log 2>&1)
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bench.csv; head -12 $O/kernel_stats_bench.csv | cut -c1-160
rm -rf $O/stats
bash tools/pmc_run.sh gpurun_out/r5b/pmc_bench bench.py --pmc-child --seconds 240
python tools/pmc_summary.py $O/pmc_bench > $O/pmc_summary.txt 2>&1
python - <<'PY' > $O/alg.env
B = 55
lv = lambda i: 8 * B * 48 * (i + 1) * (256 >> i) * (3072 >> i)
w3 = (6 * lv(0) + 6 * lv(1)) / 12
w6 = (6 * (lv(2) + lv(3) + lv(4)) + 3 * lv(5)) / 21
print(f"W3={w3:.0f}\nW6={w6:.0f}")
PY
. $O/alg.env
HOW="rocprofv3 --pmc passes (tools/pmc_run.sh) of bench.py --pmc-child --seconds 240 (one warm-up + one demix of the bench song), round-5 tree"
python tools/pmc_kernel_json.py $O/pmc_bench conv_wino3_kernel $W3 "$HOW; launches of levels 0 / 1 (48 / 96 channels)" > $O/pmc_wino3.json
python tools/pmc_kernel_json.py $O/pmc_bench conv_wino6_kernel $W6 "$HOW; launches of levels 2 .. 5 (144 .. 288 channels)" > $O/pmc_wino6.json
python tools/pmc_tdf3_json.py $O/pmc_bench --how "$HOW" > $O/pmc_tdf3.json
grep -h "traffic_over_algorithmic\|mfma_util" $O/pmc_wino3.json $O/pmc_wino6.json; python -c "
import json; d=json.load(open('$O/pmc_tdf3.json'))
for k,v in d['layers'].items(): print(k, v.get('tile'), v['dispatches'], v.get('traffic_over_algorithmic'), v.get('mfma_util'))
print(d.get('traffic_over_algorithmic_all_matched'))"
rm -rf $O/pmc_bench/*/p_agent_info.csv
du -sh $O
