#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_separate.py tests/test_gpu_ensemble.py -q -x 2>&1 | tail -3
