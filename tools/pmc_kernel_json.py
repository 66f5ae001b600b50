#!/usr/bin/env python3
"""One kernel's rocprofv3 --pmc passes (tools/pmc_run.sh) as the JSON record bench.py reads for `roofline.traffic`:

    python tools/pmc_kernel_json.py <pmc dir> <kernel name substring> <algorithmic bytes per launch> "<how it was produced>" > profiles/rNN_pmc_<kernel>.json

FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half of wide coalesced reads), WRITE_SIZE taken as reported; both are
KiB.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE summed over the XCDs x 128), as in round 2's record."""
import collections
import csv
import glob
import json
import sys

root, pat, alg, how = sys.argv[1], sys.argv[2], float(sys.argv[3]), (sys.argv[4] if len(sys.argv) > 4 else "")
agg, cnt = collections.defaultdict(float), collections.defaultdict(int)
name = None
for f in glob.glob(root + "/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            name = r["Kernel_Name"]
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
if not agg:
    sys.exit(f"no dispatch of a kernel matching {pat!r} under {root}")
per = lambda k: agg[k] / max(1, cnt[k])  # noqa: E731
fetch, write = per("FETCH_SIZE") * 1024 * 2, per("WRITE_SIZE") * 1024
print(json.dumps({
    "kernel": name.replace("void asx::", "").replace("asx::", ""), "source": how,
    "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
    "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (fetch + write) / alg,
    "mfma_busy_cycles": agg["SQ_VALU_MFMA_BUSY_CYCLES"], "grbm_gui_active_sum_over_8_xcd": agg["GRBM_GUI_ACTIVE"],
    "mfma_util": agg["SQ_VALU_MFMA_BUSY_CYCLES"] / (agg["GRBM_GUI_ACTIVE"] * 128) if agg["GRBM_GUI_ACTIVE"] else None,
    "lds_bank_conflict_cycles": agg["SQ_LDS_BANK_CONFLICT"], "lds_idx_active_cycles": agg["SQ_LDS_IDX_ACTIVE"],
    "dispatches": cnt["FETCH_SIZE"],
    "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); WRITE_SIZE as reported; counter "
            "values are KiB; per-launch figures are averages over the kernel's dispatches in the profiled command."}, indent=1))
