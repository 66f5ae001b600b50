#!/bin/bash
# round-2 kernel stats of the sibling loops (rocprofv3 --kernel-trace --stats of the probe scripts)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r3e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in roformer demucs vr; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o s -- env PYTHONPATH=$GRAFT_REPO_ROOT python $GRAFT_REPO_ROOT/tools/probe_$n.py 120 > $O/$n.log 2>&1
  head -8 $O/$n/s_kernel_stats.csv | cut -c1-120
done
