#!/bin/bash
mkdir -p gpurun_out/r3a
timeout 600 python tools/probe_host_boundary.py 2>/dev/null | tee gpurun_out/r3a/host_boundary.json
