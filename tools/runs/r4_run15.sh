#!/bin/bash
# PMC passes of the stand-alone attention harness (attention2_kernel vs attention6_kernel on the BS-Roformer shapes)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r4r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  name=$(echo "$grp" | cut -c1-20 | tr ' ' '_')
  timeout 100 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$name -o p -- $GRAFT_REPO_ROOT/tools/proto_attn6 > $O/$name.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/r4r/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void asx::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (v.get("GRBM_GUI_ACTIVE", 1) * 128) if v.get("GRBM_GUI_ACTIVE") else 0
    print(k, "mfma_busy=%.3f" % mf, "lds_conflict/active=%.3f" % (v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, v.get("SQ_LDS_IDX_ACTIVE", 1))),
          "wait_any/wave_cycles=%.3f" % (v.get("SQ_WAIT_INST_ANY", 0) / max(1.0, v.get("SQ_WAVE_CYCLES", 1))))
PY
rm -rf $O/*/p_agent_info.csv
