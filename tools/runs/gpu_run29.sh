#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_ensemble.py tests/test_gpu_separate.py -q -x 2>&1 | tail -3
