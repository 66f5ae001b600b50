#!/bin/bash
# round-2 evidence run: full GPU suite, conv KC4 A/B, bench modes, rocprofv3 stats + PMC passes of the bench configuration
set -u
O=gpurun_out/r2i
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
B="python bench.py --steps 4 --warmup 2 --cpu-seconds 0 --siblings 0"
timeout 300 $B > $O/b_base.json 2> $O/b_base.err
ASX_CONV_KC4=48 timeout 300 $B > $O/b_kc4_48.json 2> $O/b_kc4_48.err
ASX_CONV_KC4=96 timeout 300 $B > $O/b_kc4_96.json 2> $O/b_kc4_96.err
timeout 300 $B --mode chunks > $O/b_chunks.json 2> $O/b_chunks.err
timeout 300 $B --songs-per-rank 4 > $O/b_spr4.json 2> $O/b_spr4.err
BENCH_FORCE_DIST=1 timeout 300 $B --songs-per-rank 2 > $O/b_forced_dist.json 2> $O/b_forced_dist.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r2i/b_*.json')):
    try:
        txt=open(f).read().strip().splitlines()
        r=json.loads(txt[-1]); km=r['kernel_ms']; print(os.path.basename(f), len(txt), r['value'], r['ms_per_step'], r['scaling'], r['config']['songs_per_step'], {k:km.get(k) for k in ('conv3x3','tdf')}, r['rccl'])
    except Exception as e: print(f,'ERR',e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 > $GRAFT_REPO_ROOT/$O/stats.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/pmc_run.sh $O/pmc bench.py --steps 1 --warmup 1 --cpu-seconds 0 --siblings 0
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
head -30 $O/pmc_summary.txt
ls $O/stats | head; find $O/stats -name "*stats*" | head
