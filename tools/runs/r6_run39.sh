#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command on the tree of the round's second session (summary for profiles/)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6g
mkdir -p $O
cd $GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --siblings 0 --file-level 0 --cpu-seconds 0 --traffic stored --no-arith-ab > $O/stats_bench.json 2> $O/stats_bench.err)
f=$(find $O/stats -name "s_kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats_bench_s2.csv
head -12 $O/kernel_stats_bench_s2.csv | cut -c1-200
tail -1 $O/stats_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launches'])"
