#!/bin/bash
# per-launch times of the 3x3 class (ASX_PROF_DUMP) for the stationary kernel and for conv_wino3_kernel, same box
set -u
O=gpurun_out/r4d
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
for w in 1 0; do
ASX_WINOS=$w ASX_PROF_DUMP=1 timeout 600 python bench.py --gpus 1 --steps 1 --warmup 1 --cpu-seconds 0 --siblings 0 --file-level 0 2> $O/prof_dump$w.err > /dev/null
grep "cls=1 " $O/prof_dump$w.err | head -33 | awk '{print $3}' | tr '\n' ' '; echo
done
