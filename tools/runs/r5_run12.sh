#!/bin/bash
# frames-per-workgroup sweep of the 6144-point STFT / iSTFT kernels on the final tree (ASX_FFT3_GS / ASX_FFT3_G), one call
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r5l
mkdir -p $O
cd $GRAFT_REPO_ROOT
for g in 6 8 12 16 24 32 64; do
  ASX_FFT3_GS=$g ASX_FFT3_G=$g timeout 120 python bench.py --steps 3 --warmup 2 --cpu-seconds 0 --siblings 0 --file-level 0 --traffic stored > $O/b_$g.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/b_$g.json')); s=d['stage_roofline']; print('G=GS=$g', 'stft', d['kernel_ms']['stft'], s['stft']['frac'], 'istft', d['kernel_ms']['istft'], s['istft']['frac'])" | tee -a $O/fft_sweep.txt
done
