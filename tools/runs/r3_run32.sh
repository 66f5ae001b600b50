#!/bin/bash
# PMC passes of the bench configuration with the Winograd kernel
set -u
O=gpurun_out/r3w
export PYTHONPATH=$GRAFT_REPO_ROOT
bash tools/pmc_run.sh $O/pmc_bench bench.py --steps 1 --warmup 1 --cpu-seconds 0 --siblings 0 --file-level 0
python tools/pmc_summary.py $O/pmc_bench > $O/pmc_bench_summary.txt 2>&1
head -30 $O/pmc_bench_summary.txt
python tools/pmc_kernel_json.py $O/pmc_bench conv_wino3_kernel 5352652800 "rocprofv3 --pmc passes of bench.py --steps 1 --warmup 1 (tools/pmc_run.sh), round 3, Winograd default" > $O/r03_pmc_wino3.json
cat $O/r03_pmc_wino3.json
