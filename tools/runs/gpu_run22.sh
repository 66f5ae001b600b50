#!/bin/bash
set -u
O=gpurun_out/r2v
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "fft3" 2>&1 | tail -3
timeout 600 python tools/probe_two_streams.py 120 2>$O/two.err | tee $O/two.json
tail -3 $O/two.err
