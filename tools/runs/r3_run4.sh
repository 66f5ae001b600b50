#!/bin/bash
# round-3 GPU run 4: single-barrier attention (A/B), halo grid threshold (A/B), rocprofv3 stats of the siblings on this tree, PMC of the halo kernel
set -u
O=gpurun_out/r3d
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_demucs.py tests/test_gpu_hdemucs.py tests/test_gpu_roformer.py tests/test_gpu_fullsize.py -q -x -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
S="python tools/bench_siblings.py --cpu 0 --steps 2"
timeout 600 $S --workloads htdemucs,hdemucs,roformer,vr > $O/sib_default.jsonl 2> $O/sib_default.err
ASX_MHA_DB=0 timeout 400 $S --workloads htdemucs,hdemucs > $O/sib_mhadb0.jsonl 2> $O/sib_mhadb0.err
ASX_ATTN_DB=1 timeout 400 $S --workloads roformer > $O/sib_attndb1.jsonl 2> $O/sib_attndb1.err
ASX_HALO_MINBLK=600 timeout 600 $S --workloads vr,htdemucs,hdemucs > $O/sib_minblk600.jsonl 2> $O/sib_minblk600.err
ASX_HALO_MINBLK=1600 timeout 600 $S --workloads vr,htdemucs,hdemucs > $O/sib_minblk1600.jsonl 2> $O/sib_minblk1600.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r3d/sib_*.jsonl')):
    for l in open(f):
        try:
            r=json.loads(l); print(os.path.basename(f), r['config']['workload'][:18], r['value'], r['ms_per_step'], r['roofline']['frac'], {k[:22]:v for k,v in (r.get('kernel_ms') or {}).items() if v>5})
        except Exception as e: print(f,'ERR',e, l[:100])
PY
ASX_PROF_DUMP=1 timeout 300 python tools/probe_hdemucs.py 120 4 2 > $O/dump_hd.log 2> $O/dump_hd.err
tail -12 $O/dump_hd.log
cd /tmp && export TMPDIR=/tmp
for w in demucs roformer vr hdemucs; do
  args="60"; [ $w = demucs ] && args="60 8 2"; [ $w = hdemucs ] && args="120 4 2"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_$w -o s -- python $GRAFT_REPO_ROOT/tools/probe_$w.py $args > $GRAFT_REPO_ROOT/$O/stats_$w.log 2>&1
done
cd $GRAFT_REPO_ROOT
bash tools/pmc_run.sh $O/pmc_ht tools/probe_demucs.py 30 8 2
python tools/pmc_summary.py $O/pmc_ht > $O/pmc_ht_summary.txt 2>&1
grep -A10 "hg_kernel" $O/pmc_ht_summary.txt | head -50
find $O -name "*kernel_stats.csv" | head
