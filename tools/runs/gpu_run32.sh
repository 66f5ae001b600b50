#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_gpu_roformer.py tests/test_gpu_fullsize.py tests/test_gpu_separate.py -q -x 2>&1 | tail -3
for q in 2 1; do
  ASX_ATTN_QW=$q timeout 600 python tools/probe_roformer.py 240 8 2>/dev/null | grep -E "audio|attention" | sed "s/^/QW=$q /"
done
