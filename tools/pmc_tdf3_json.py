#!/usr/bin/env python3
"""Per-LAYER traffic record of the split-operand row GEMM (tdf3_kernel, bf16 x 6 or fp16 x 3: the record says which ran) from rocprofv3 --pmc passes of the bench song (tools/pmc_run.sh):
round 4's record averaged every TDF dispatch against `algorithmic_bytes_per_launch: 1.0` ("not filled in").
Here every dispatch is matched to its layer by its grid -- ceil(M / BM) x ceil(N / BN) workgroups of 256 threads, BM / BN from the
kernel's template arguments -- and compared with THAT layer's algorithmic bytes: x [M, K] + y [M, N] (+ the residual [M, N] on the
second linear of a TDF block) + the fp32 weights [N, K], M = chunks x C x T (uvr_lib_v5/modules.py:57-74).

    python tools/pmc_tdf3_json.py <pmc dir> [--chunks 55] [--g 48] [--dim-f 3072] [--dim-t 256] [--levels 6] [--bn 8] > profiles/r05_pmc_tdf3.json

FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half of wide coalesced reads), WRITE_SIZE as reported; both KiB.
FETCH_SIZE counts L2 misses towards the fabric: Infinity-Cache hits are included, so `traffic` is fabric traffic, an upper bound
of HBM traffic."""
import argparse
import collections
import csv
import glob
import json
import re
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--chunks", type=int, default=55)
    ap.add_argument("--g", type=int, default=48)
    ap.add_argument("--dim-f", type=int, default=3072)
    ap.add_argument("--dim-t", type=int, default=256)
    ap.add_argument("--levels", type=int, default=6)
    ap.add_argument("--bn", type=int, default=8)
    ap.add_argument("--how", default="")
    a = ap.parse_args()
    layers = []                                         # (name, M, N, K, algorithmic bytes)
    for lv in range(a.levels):
        c, t, f = a.g * (lv + 1), a.dim_t >> lv, a.dim_f >> lv
        m = a.chunks * c * t
        layers.append((f"L{lv}.F_to_F8", m, f // a.bn, f, 4.0 * (m * f + m * (f // a.bn) + (f // a.bn) * f)))
        layers.append((f"L{lv}.F8_to_F", m, f, f // a.bn, 4.0 * (m * (f // a.bn) + 2 * m * f + f * (f // a.bn))))
    # per dispatch: counters keyed by (Dispatch_Id) do not line up across passes; aggregate by (kernel template, grid) instead
    arith = set()
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for fn in glob.glob(a.root + "/*/p_counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if "tdf3_kernel" not in k:
                continue
            mt = re.search(r"tdf3_kernel<(\d+), (\d+), (\d+), (\w+)(?:, (\w+))?>", k)   # NREP, MREP, ABL, GATHER[, H]
            if not mt or mt.group(4) not in ("false", "0"):
                continue
            if mt.group(5) in ("true", "1"):
                arith.add("fp16 x 3")
            else:
                arith.add("bf16 x 6")
            nrep, mrep = int(mt.group(1)), int(mt.group(2))
            wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
            key = (nrep, mrep, wgs)
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[key][r["Counter_Name"]] += 1
    if not agg:
        sys.exit(f"no tdf3_kernel dispatch under {a.root}")
    out = {"kernel": "tdf3_kernel<NREP, MREP, 0, false[, H]> (split-operand row GEMM, csrc/kernels_gemm3.h)", "arithmetic": sorted(arith), "source": a.how, "layers": {},
           "note": "per layer of the HQ_3 net (55 chunks per launch): FETCH_SIZE x 2 + WRITE_SIZE per dispatch (KiB counters; gfx950 correction of "
                   "MI355X_MICROARCH.md) against the layer's algorithmic bytes x + y (+ residual) + W; FETCH_SIZE counts L2 misses towards the "
                   "fabric, Infinity-Cache hits included"}
    tot_t, tot_a = 0.0, 0.0
    for (nrep, mrep, wgs), v in sorted(agg.items(), key=lambda kv: -kv[0][2]):
        bm, bn = 16 * mrep, 64 * nrep
        match = [L for L in layers if -(-L[1] // bm) * -(-L[2] // bn) == wgs]
        n = max(1, cnt[(nrep, mrep, wgs)]["FETCH_SIZE"])
        fetch = v["FETCH_SIZE"] / n * 1024 * 2
        write = v["WRITE_SIZE"] / max(1, cnt[(nrep, mrep, wgs)]["WRITE_SIZE"]) * 1024
        rec = {"tile": f"{bm} x {bn}", "workgroups": wgs, "dispatches": n, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
               "traffic_bytes_per_launch": fetch + write}
        if v.get("GRBM_GUI_ACTIVE"):
            rec["mfma_util"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] * 128), 4)
        if v.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_conflict_over_active"] = round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 4)
        if len(match) == 1:
            name, m, nn, k, alg = match[0]
            rec.update({"M": m, "N": nn, "K": k, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round((fetch + write) / alg, 4)})
            tot_t += (fetch + write) * n
            tot_a += alg * n
            out["layers"][name] = rec
        else:
            out["layers"][f"unmatched_{nrep}x{mrep}_{wgs}"] = dict(rec, candidates=[L[0] for L in match])
    if tot_a:
        out["traffic_over_algorithmic_all_matched"] = round(tot_t / tot_a, 4)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
