#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_gpu_mdxc.py tests/test_gpu_parity.py -q -x -k "mdxc or conv or v3" 2>&1 | tail -3
for k in 1 0; do
  ASX_CONV_KC4_N2=$k timeout 600 python tools/probe_mdxc.py 120 2>/dev/null | grep -E "audio|conv3x3|tdf|misc" | sed "s/^/KC4_N2=$k /"
done
