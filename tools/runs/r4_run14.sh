#!/bin/bash
# final round-4 evidence: whole GPU suite, smoke(), the driver's bench command, whole-song parity of every BASELINE config
# against the oracle records (gpurun_cache/fullsong, computed on the CPU beforehand), rocprofv3 kernel stats of the bench
# command and of the Roformer sibling, PMC passes of the 55-chunk configuration, the stand-alone harnesses
set -u
O=gpurun_out/r4n
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r4n/bench_n1.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic'], r.get('parity_rel_rms_vs_cpu'))
print(r['kernel_ms'])
print(r['stage_roofline']['tdf'])
print({k:v.get('value') for k,v in r['siblings'].items()}, r['file_level'].get('rtf'))
PY
if [ -d gpurun_cache/fullsong ]; then
  timeout 900 python tools/fullsong_parity.py > $O/fullsong_parity.json 2> $O/fullsong_parity.err
  python -c "
import json
r=json.load(open('gpurun_out/r4n/fullsong_parity.json'))
print({k:(v.get('worst_rel_rms'), v.get('pass')) for k,v in r['cases'].items()})"
fi
for nt in 0 6; do ASX_NT=$nt timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --siblings 0 --file-level 0 --traffic stored 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ASX_NT=$nt', r['value'], r['kernel_ms']['tdf'])"; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_bench -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0 --traffic stored > $GRAFT_REPO_ROOT/$O/stats_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_roformer -o s -- python $GRAFT_REPO_ROOT/tools/bench_siblings.py --workloads roformer --cpu 0 --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$O/stats_roformer.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_vr -o s -- python $GRAFT_REPO_ROOT/tools/bench_siblings.py --workloads vr,htdemucs --cpu 0 --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$O/stats_vr_ht.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/pmc_run.sh $O/pmc_bench bench.py --steps 1 --warmup 1 --cpu-seconds 0 --siblings 0 --file-level 0 --traffic stored
python tools/pmc_summary.py $O/pmc_bench > $O/pmc_bench_summary.txt 2>&1
python tools/pmc_kernel_json.py $O/pmc_bench conv_wino3_kernel 5352652800 "rocprofv3 --pmc passes of bench.py --steps 1 --warmup 1 --traffic stored (tools/pmc_run.sh), final round-4 tree" > $O/pmc_wino3.json
python tools/pmc_kernel_json.py $O/pmc_bench "tdf3_kernel<3, 8" 1 "rocprofv3 --pmc passes of the same command: the bf16x6 row GEMM (all TDF launches averaged; algorithmic bytes not filled in)" > $O/pmc_tdf3.json
grep -A3 '"mfma_util"\|lds_bank' $O/pmc_tdf3.json | head -8
timeout 200 tools/proto_gemm3 0 > $O/proto_gemm3.txt 2>&1
timeout 200 tools/proto_attn6 > $O/proto_attn6.txt 2>&1
find $O -name "*kernel_stats.csv" | head -5
rm -rf $O/pmc_bench/*/p_agent_info.csv
