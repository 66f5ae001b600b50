#!/bin/bash
# round 6, run 3: PMC passes of conv3h_kernel at the level-0 shape (traffic, L2 hit rate, LDS, MFMA busy)
mkdir -p gpurun_out/r6a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o 'TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*' | sort -u | tr '\n' ' ' > gpurun_out/r6a/counters.txt
bash tools/pmc_bin.sh gpurun_out/r6a/pmc_conv3h tools/proto_conv3h 0 2 55 3 1
python3 tools/pmc_bin_summary.py gpurun_out/r6a/pmc_conv3h conv3h > gpurun_out/r6a/pmc_conv3h.txt 2>&1
cat gpurun_out/r6a/pmc_conv3h.txt
tail -5 gpurun_out/r6a/pmc_conv3h/*.log | tail -30
