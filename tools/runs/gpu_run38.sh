#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "capturable" 2>&1 | tail -15
