#!/bin/bash
# Winograd third generation: ablations
mkdir -p gpurun_out/r3w
for ab in 0 1 2 4 8 16 15; do WINO=3 ASX_WINO_ABL=$ab timeout 200 python tools/probe_wino.py 2>&1 | grep WINO | cut -c1-160 | tee -a gpurun_out/r3w/launches4.log; done
