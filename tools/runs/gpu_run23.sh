#!/bin/bash
# rotary in the qkv projection's epilogue: parity (both settings) + A/B on the ep_317 layout
set -u
O=gpurun_out/r2w
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_roformer.py tests/test_gpu_fullsize.py tests/test_gpu_separate.py -q -x 2>&1 | tail -3
ASX_ROF_FUSE=0 timeout 600 python -m pytest tests/test_gpu_roformer.py -q -x 2>&1 | tail -2
for f in 1 0; do
  ASX_ROF_FUSE=$f timeout 600 python tools/probe_roformer.py 240 8 2>/dev/null | grep -E "audio|gemm|attention|misc" | sed "s/^/FUSE=$f /"
done
