#!/bin/bash
# round-3 GPU run 7: batch-size sweeps of the sibling loops (engine knobs; results do not depend on them), VR tests after the batching change
set -u
O=gpurun_out/r3g
mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_vr.py -q -x -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for mb in 28 42 84; do
  ASX_HALO_MINBLK=600 timeout 300 python tools/probe_vr.py 240 $mb > $O/vr_mb$mb.log 2>&1; echo "vr mb=$mb"; grep "audio\|kernel ms" $O/vr_mb$mb.log
done
timeout 300 python tools/probe_vr.py 240 28 > $O/vr_mb28_nominblk.log 2>&1; echo "vr mb=28 minblk=0"; grep "audio\|kernel ms" $O/vr_mb28_nominblk.log
for mb in 14 21 28 42; do
  timeout 300 python tools/probe_demucs.py 240 $mb 2 > $O/ht_mb$mb.log 2>&1; echo "htdemucs mb=$mb"; grep "^audio" $O/ht_mb$mb.log
done
for mb in 8 16 30; do
  timeout 400 python tools/probe_roformer.py 240 $mb > $O/rof_mb$mb.log 2>&1; echo "roformer mb=$mb"; grep "^audio" $O/rof_mb$mb.log
done
for mb in 8 16 24; do
  timeout 300 python tools/probe_hdemucs.py 240 $mb 2 > $O/hd_mb$mb.log 2>&1; echo "hdemucs mb=$mb"; grep "^audio" $O/hd_mb$mb.log
done
