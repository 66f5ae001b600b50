#!/bin/bash
# evidence on the bf16x6 row-GEMM tree: whole GPU suite, smoke(), the driver's bench command, rocprofv3 kernel stats of the bench
# command, PMC passes (MFMA busy / LDS / traffic) of the 55-chunk configuration
set -u
O=gpurun_out/r4m
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r4m/bench_n1.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic'], r.get('parity_rel_rms_vs_cpu'))
print(r['kernel_ms'])
print(r['stage_roofline']['tdf'])
print({k:v.get('value') for k,v in r['siblings'].items()}, r['file_level'].get('rtf'))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_bench -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0 --traffic stored > $GRAFT_REPO_ROOT/$O/stats_bench.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/pmc_run.sh $O/pmc_bench bench.py --steps 1 --warmup 1 --cpu-seconds 0 --siblings 0 --file-level 0 --traffic stored
python tools/pmc_summary.py $O/pmc_bench > $O/pmc_bench_summary.txt 2>&1
python tools/pmc_kernel_json.py $O/pmc_bench conv_wino3_kernel 5352652800 "rocprofv3 --pmc passes of bench.py --steps 1 --warmup 1 --traffic stored (tools/pmc_run.sh), round 4 tree with the bf16x6 row GEMM" > $O/r04b_pmc_wino3.json
python tools/pmc_kernel_json.py $O/pmc_bench "tdf3_kernel<3, 8" 1 "rocprofv3 --pmc passes of the same command: the bf16x6 row GEMM (all TDF launches averaged; algorithmic bytes not filled in)" > $O/r04b_pmc_tdf3.json
head -12 $O/pmc_bench_summary.txt
cat $O/r04b_pmc_tdf3.json | head -20
find $O -name "*kernel_stats.csv" | head -3
rm -rf $O/pmc_bench/*/p_agent_info.csv
