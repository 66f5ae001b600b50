#!/bin/bash
# round-3 evidence run (final tree, after the occupancy changes): whole GPU suite, the default bench line (driver's command), rocprofv3 stats of the bench and of the sibling probes,
# PMC passes of the bench configuration (3x3 conv) and of the VR probe (halo kernel), RCCL code path with one rank
set -u
O=gpurun_out/r3y
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r3y/bench_n1.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['cpu_baseline'], r.get('parity_rel_rms_vs_cpu'))
print({k:v.get('value') for k,v in r['siblings'].items()}, r['file_level']['rtf'] if 'rtf' in r['file_level'] else r['file_level'])
print({k:(v['frac'],v['bound']) for k,v in r['stage_roofline'].items()})
PY
B="python bench.py --steps 3 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0"
BENCH_FORCE_DIST=1 timeout 300 $B --config5 --songs-per-rank 2 > $O/b_forced_files.json 2> $O/b_forced_files.err
BENCH_FORCE_DIST=1 timeout 300 $B --mode chunks > $O/b_forced_chunks.json 2> $O/b_forced_chunks.err
python - <<'PY'
import json
for f in ('b_forced_files','b_forced_chunks'):
    try:
        r=json.loads(open(f'gpurun_out/r3y/{f}.json').read().strip().splitlines()[-1]); print(f, r['value'], r['ms_per_step'], r['scaling'], r['rccl'], r['config']['workload'][-120:])
    except Exception as e: print(f,'ERR',e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_bench -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0 > $GRAFT_REPO_ROOT/$O/stats_bench.log 2>&1
for w in demucs roformer vr hdemucs; do
  args="240"; [ $w = demucs ] && args="240 32 2"; [ $w = hdemucs ] && args="240 16 2"; [ $w = vr ] && args="240 42"; [ $w = roformer ] && args="60 8"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_$w -o s -- python $GRAFT_REPO_ROOT/tools/probe_$w.py $args > $GRAFT_REPO_ROOT/$O/stats_$w.log 2>&1
done
cd $GRAFT_REPO_ROOT
bash tools/pmc_run.sh $O/pmc_bench bench.py --steps 1 --warmup 1 --cpu-seconds 0 --siblings 0 --file-level 0
python tools/pmc_summary.py $O/pmc_bench > $O/pmc_bench_summary.txt 2>&1
bash tools/pmc_run.sh $O/pmc_ht tools/probe_demucs.py 60 16 2
python tools/pmc_summary.py $O/pmc_ht > $O/pmc_ht_summary.txt 2>&1
bash tools/pmc_run.sh $O/pmc_vr tools/probe_vr.py 60 21
python tools/pmc_summary.py $O/pmc_vr > $O/pmc_vr_summary.txt 2>&1
head -14 $O/pmc_bench_summary.txt; grep -A11 "hg_kernel" $O/pmc_vr_summary.txt | head -40
find $O -name "*kernel_stats.csv"
