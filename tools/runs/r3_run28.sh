#!/bin/bash
# Winograd v2: per-launch times and ablations
mkdir -p gpurun_out/r3w
for w in 0 2; do WINO=$w timeout 200 python tools/probe_wino.py 2>&1 | grep WINO | tee -a gpurun_out/r3w/launches.log; done
for ab in 1 2 4 8 3 11 15; do WINO=2 ASX_WINO_ABL=$ab timeout 200 python tools/probe_wino.py 2>&1 | grep WINO | tee -a gpurun_out/r3w/launches.log; done
