#!/bin/bash
# rocprofv3 PMC passes (one counter group per run, no tracing) of a probe; usage: tools/pmc_run.sh <outdir> <probe args...>
# afterwards: python tools/pmc_summary.py <outdir>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; shift
mkdir -p "$OUT"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  name=$(echo "$grp" | cut -c1-24 | tr ' ' '_')
  (cd /tmp && export TMPDIR=/tmp && PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/$name" -o p -- python "$R/$1" "${@:2}" > "$OUT/$name.log" 2>&1)
done
