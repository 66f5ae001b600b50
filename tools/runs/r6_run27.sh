#!/bin/bash
# round 6, run 27: what the chip reports while conv3h_kernel loops (power cap, socket power, shader / memory clocks), against the compute-only ablation and an idle chip
mkdir -p gpurun_out/r6e
cd $GRAFT_REPO_ROOT
{
echo "== idle"; rocm-smi --showpower --showclocks --showmaxpower --showperflevel 2>&1 | grep -v '^$' | head -30
for abl in 0 6; do
  echo "== harness abl $abl looping (level-0 shape, 200 launches)"
  timeout 120 tools/proto_conv3h $abl 2 55 200 1 > gpurun_out/r6e/loop_$abl.txt 2>&1 &
  sleep 6
  for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -E 'Power|sclk|mclk|fclk|socclk' | head -8; sleep 1; done
  wait
  tail -2 gpurun_out/r6e/loop_$abl.txt | head -1
done
} > gpurun_out/r6e/power_state.txt 2>&1
cat gpurun_out/r6e/power_state.txt
