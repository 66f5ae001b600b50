#!/bin/bash
# round-3 GPU run 9: whole GPU suite on the tree with the Demucs / VR device-resident file paths, Demucs file-level rate
set -u
O=gpurun_out/r3i
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
timeout 600 python tools/probe_file_level.py > $O/file_level_htdemucs.json 2> $O/file_level_htdemucs.err
cat $O/file_level_htdemucs.json; tail -3 $O/file_level_htdemucs.err
