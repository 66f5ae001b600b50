#!/bin/bash
# round-3 evidence run on the final tree (Winograd default): whole GPU suite, the driver's bench command, ring-depth A/B on the same box,
# RCCL code path with one rank, rocprofv3 stats + PMC passes of the bench configuration, whole-song parity of every case
set -u
O=gpurun_out/r3v
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -2
tail -4 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r3v/bench_n1.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline'].get('executed'), r['roofline'].get('direct_kernel'), r['cpu_baseline'], r.get('parity_rel_rms_vs_cpu'))
print({k:v.get('value') for k,v in r['siblings'].items()}, r['file_level']['rtf'] if 'rtf' in r['file_level'] else r['file_level'])
print({k:(v['frac'],v['bound']) for k,v in r['stage_roofline'].items()})
PY
B="python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0"
for c in 6 5 6 5; do ASX_WINO_CFG=$c timeout 300 $B 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CFG=$c', r['value'], r['ms_per_step'], r['kernel_ms']['conv3x3'])"; done | tee $O/ring_ab.txt
ASX_WINOGRAD=0 timeout 300 $B > $O/bench_direct.json 2>/dev/null
BENCH_FORCE_DIST=1 timeout 300 $B --config5 --songs-per-rank 2 > $O/b_forced_files.json 2> $O/b_forced_files.err
BENCH_FORCE_DIST=1 timeout 300 $B --mode chunks > $O/b_forced_chunks.json 2> $O/b_forced_chunks.err
python - <<'PY'
import json
for f in ('bench_direct','b_forced_files','b_forced_chunks'):
    try:
        r=json.loads(open(f'gpurun_out/r3v/{f}.json').read().strip().splitlines()[-1]); print(f, r['value'], r['ms_per_step'], r['scaling'], r['rccl'], r['config']['workload'][-100:])
    except Exception as e: print(f,'ERR',e)
PY
timeout 900 python tools/fullsong_parity.py > $O/fullsong_parity.json 2> $O/fullsong_parity.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3v/fullsong_parity.json'))
for k,v in d.get('cases',{}).items():
    print(k, {s:float('%.3g'%x['rel_rms']) for s,x in v['stems'].items()})
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_bench -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0 > $GRAFT_REPO_ROOT/$O/stats_bench.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/pmc_run.sh $O/pmc_bench bench.py --steps 1 --warmup 1 --cpu-seconds 0 --siblings 0 --file-level 0
python tools/pmc_summary.py $O/pmc_bench > $O/pmc_bench_summary.txt 2>&1
python tools/pmc_kernel_json.py $O/pmc_bench conv_wino3_kernel 5352652800 "rocprofv3 --pmc passes of bench.py --steps 1 --warmup 1 (tools/pmc_run.sh), round 3 final tree" > $O/r03_pmc_wino3.json
python tools/pmc_kernel_json.py $O/pmc_bench "conv_dma_kernel<asx::ConvDmaCfg<3, 3, 1, 1, 3, 4" 5352652800 "rocprofv3 --pmc passes of bench.py --steps 1 --warmup 1 (tools/pmc_run.sh), round 3 final tree; the direct kernel runs in the bench's direct_kernel leg" > $O/r03_pmc_conv3x3.json
head -12 $O/pmc_bench_summary.txt
find $O -name "*kernel_stats.csv"
