#!/bin/bash
# what the chip sustains on a pure stream of matrix instructions (tools/experimental/micro_mfma.hip), and at what clock (GRBM_GUI_ACTIVE per dispatch)
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r6h
mkdir -p $O; rm -rf $O/mfma_clock
cd $GRAFT_REPO_ROOT
tools/experimental/micro_mfma 2>&1 | grep -v amdgpu.ids | tee $O/micro_mfma.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/mfma_clock -o p -- $GRAFT_REPO_ROOT/tools/experimental/micro_mfma > /dev/null 2>&1)
python3 - <<'PY' | tee -a gpurun_out/r6h/micro_mfma.txt
import csv, glob
root = "gpurun_out/r6h/mfma_clock"
cyc, dur, name = {}, {}, {}
for f in glob.glob(root + "/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc[r["Dispatch_Id"]] = float(r["Counter_Value"]); name[r["Dispatch_Id"]] = r["Kernel_Name"]
for f in glob.glob(root + "/**/p_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]))
print("# clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch time, timed dispatches only")
for d in sorted(cyc, key=int):
    if d in dur and dur[d][0] > 2e6:
        print(f"dispatch {int(d):3d} {name[d][:28]:28s} grid {dur[d][1]:7d}  {dur[d][0] / 1e6:8.3f} ms  clock {cyc[d] / dur[d][0] / 8:.3f} GHz")
PY
