#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_roformer.py tests/test_gpu_parity.py tests/test_gpu_separate.py -q -x 2>&1 | tail -3
