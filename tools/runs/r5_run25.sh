#!/bin/bash
# kernel tables of the sibling loops on the final tree (fp16 x 3 row GEMM / GATHER convs / attention): htdemucs, VR, BS-Roformer, hdemucs_mmi
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5z
mkdir -p $O
cd $GRAFT_REPO_ROOT
prof() {  # name, probe args...
  local name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && PYTHONPATH=$GRAFT_REPO_ROOT timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$name -o s -- python "$@" > $O/$name.log 2>&1)
  find $O/st_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$name.csv
  rm -rf $O/st_$name
  echo "== $name"; tail -2 $O/$name.log; head -14 $O/kernel_stats_$name.csv | cut -c1-150
}
prof htdemucs $GRAFT_REPO_ROOT/tools/probe_demucs.py 60 28 2
prof vr $GRAFT_REPO_ROOT/tools/probe_vr.py 60 48
prof roformer $GRAFT_REPO_ROOT/tools/probe_roformer.py 64 16
prof hdemucs $GRAFT_REPO_ROOT/tools/probe_hdemucs.py 60 4 2
