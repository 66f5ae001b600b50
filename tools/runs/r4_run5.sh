#!/bin/bash
# free-running-waves Winograd kernel (conv_winof_kernel): parity, per-launch A/B against the other two, ablations
set -u
O=gpurun_out/r4e
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "winograd_stationary" > $O/pytest_wino.log 2>&1; echo "rc=$?" >> $O/pytest_wino.log
tail -5 $O/pytest_wino.log
for w in 2 1 0; do
ASX_WINOS=$w ASX_PROF_DUMP=1 timeout 600 python bench.py --gpus 1 --steps 1 --warmup 1 --cpu-seconds 0 --siblings 0 --file-level 0 2> $O/prof_dump$w.err > $O/bench$w.json
echo "WINOS=$w"; grep "cls=1 " $O/prof_dump$w.err | head -33 | awk '{print $3}' | tr '\n' ' '; echo
done
for abl in 1 2 8 9; do
ASX_WINOS=2 ASX_WINOF_ABL=$abl ASX_PROF_DUMP=1 timeout 600 python bench.py --gpus 1 --steps 1 --warmup 1 --cpu-seconds 0 --siblings 0 --file-level 0 2> $O/prof_abl$abl.err > /dev/null
echo "WINOF_ABL=$abl"; grep "cls=1 " $O/prof_abl$abl.err | head -33 | awk '{print $3}' | tr '\n' ' '; echo
done
