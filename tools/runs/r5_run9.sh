#!/bin/bash
# Round-5 evidence on the final tree: the driver's three commands (pytest -m gpu -x, smoke, bench), rocprofv3 kernel stats of the bench,
# PMC passes (conv_wino3 / conv_wino6 / tdf3 per layer), single-rank RCCL records of both bench modes, and the level-1 A/B of the
# bf16 x 6 Winograd threshold inside the net.
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5i
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5i/bench_n1.json"))
print(d["value"], d["ms_per_step"], d["kernel_ms"])
r = d["roofline"]; print({k: r[k] for k in ("achieved", "frac", "traffic", "share_of_step_ms", "conv3x3_class_ms")})
for k, v in r["per_level"]["conv3x3"].items(): print(k, v["kernel"][:18], v["avg_launch_ms"], v["frac"])
print({k: (v.get("value"), v.get("ms_per_step")) for k, v in d.get("siblings", {}).items()})
print(d.get("file_level", {}).get("rtf"), d.get("parity_rel_rms_vs_cpu"), d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
PY
for w6 in 96 144 0; do
  ASX_WINO6=$w6 timeout 200 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --siblings 0 --file-level 0 --traffic stored > $O/bench_wino6_$w6.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/bench_wino6_$w6.json')); print('ASX_WINO6=$w6', d['value'], d['ms_per_step'], d['kernel_ms']['conv3x3'], [(k, v['avg_launch_ms']) for k, v in d['roofline']['per_level']['conv3x3'].items()])"
done
BENCH_FORCE_DIST=1 timeout 200 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0 --traffic stored --mode files > $O/bench_force_dist_files.json 2> $O/fd_files.err; echo "force-dist files rc=$?"
BENCH_FORCE_DIST=1 timeout 200 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0 --traffic stored --mode chunks > $O/bench_force_dist_chunks.json 2> $O/fd_chunks.err; echo "force-dist chunks rc=$?"
python -c "
import json
for m in ('files','chunks'):
    d=json.load(open('$O/bench_force_dist_%s.json' % m)); print(m, d['value'], d['ms_per_step'], d['rccl'])"
(cd /tmp && export TMPDIR=/tmp && PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0 --traffic stored > $O/stats.log 2>&1)
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_bench.csv; head -8 $O/kernel_stats_bench.csv | cut -c1-150
rm -rf $O/stats
bash tools/pmc_run.sh gpurun_out/r5i/pmc_bench bench.py --pmc-child --seconds 240
python tools/pmc_summary.py $O/pmc_bench > $O/pmc_summary.txt 2>&1
HOW="rocprofv3 --pmc passes (tools/pmc_run.sh) of bench.py --pmc-child --seconds 240 (one warm-up + one demix of the bench song), final round-5 tree"
python tools/pmc_kernel_json.py $O/pmc_bench conv_wino3_kernel 12457082880 "$HOW; launches of levels 0 / 1 (48 / 96 channels)" > $O/pmc_wino3.json
python tools/pmc_kernel_json.py $O/pmc_bench conv_wino6_kernel 1292977006 "$HOW; launches of levels 2 .. 5 (144 .. 288 channels)" > $O/pmc_wino6.json
python tools/pmc_tdf3_json.py $O/pmc_bench --how "$HOW" > $O/pmc_tdf3.json
grep -h "traffic_over_algorithmic\|mfma_util" $O/pmc_wino3.json $O/pmc_wino6.json
rm -rf $O/pmc_bench/*/p_agent_info.csv $O/pmc_bench/*/p_kernel_trace.csv
du -sh $O
