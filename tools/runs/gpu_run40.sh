#!/bin/bash
# Roformer token rows padded to a multiple of 8 (second-generation row GEMM for every batch size)
set -u
timeout 600 python -m pytest tests/test_gpu_roformer.py tests/test_gpu_fullsize.py tests/test_gpu_separate.py -q -x 2>&1 | tail -3
timeout 300 python tools/probe_roformer.py 240 8 2>/dev/null | grep -E "audio|gemm"
