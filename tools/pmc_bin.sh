#!/bin/bash
# rocprofv3 PMC passes (one counter group per run, kernel trace only) of a stand-alone harness binary;
# usage: tools/pmc_bin.sh <outdir> <binary> <args...>   ->   <outdir>/<group>/p_counter_collection.csv
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; shift
mkdir -p "$OUT"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  name=$(echo "$grp" | cut -c1-24 | tr ' ' '_')
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/$name" -o p -- "$R/$1" "${@:2}" > "$OUT/$name.log" 2>&1)
done
