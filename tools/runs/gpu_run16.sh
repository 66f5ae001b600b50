#!/bin/bash
set -u
O=gpurun_out/r2p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ASX_FFT3_GS=16 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --siblings 0 > $GRAFT_REPO_ROOT/$O/stats_bench.json 2> $GRAFT_REPO_ROOT/$O/stats.log
cd $GRAFT_REPO_ROOT
grep -i "f3::\|chunk_table\|finalize\|fold\|ola" $O/stats/s_kernel_stats.csv | cut -c1-200
