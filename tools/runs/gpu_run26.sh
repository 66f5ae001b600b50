#!/bin/bash
# RMSNorm folded into the Roformer projections: parity (all settings) + A/B on the ep_317 layout
set -u
O=gpurun_out/r2z
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_roformer.py tests/test_gpu_fullsize.py tests/test_gpu_separate.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3
ASX_ROF_NORMFUSE=0 timeout 600 python -m pytest tests/test_gpu_roformer.py -q -x 2>&1 | tail -2
for f in 1 0; do
  ASX_ROF_NORMFUSE=$f timeout 600 python tools/probe_roformer.py 240 8 2>/dev/null | grep -E "audio|gemm|attention|misc" | sed "s/^/NORMFUSE=$f /"
done
