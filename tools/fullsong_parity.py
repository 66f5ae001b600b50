#!/usr/bin/env python3
"""GPU leg of the whole-song parity records (one per BASELINE config): the HIP engine on the same seeded inputs as
tools/fullsong_oracle.py (tools/fullsong_gpu.py), compared with the oracle's stored output at the comparison windows of
tools/fullsong_cases.py.

    python tools/fullsong_parity.py [--cases mdx_hq3,htdemucs,hdemucs_mmi,vr_2hp,vr_2hp_sinc,mdx23c,bs_roformer] > profiles/r05_fullsong_parity.json

A case whose oracle record (gpurun_cache/fullsong/<case>.npz) is missing is computed on the spot with the oracle (slow: the
CPU leg of a 4-minute htdemucs song is minutes).  Reported per array: relative RMS error (the north-star metric, bar 1e-4),
ABSOLUTE RMS error and the reference's RMS in the windows -- the synthetic nets are scaled so that stems are O(0.1), i.e. the 0.9
normalisation threshold, the 1e-6 silence threshold and the int16 quantisation run in their real range; for the MDX case the
writer's int16 stream is compared too (LSB differences).  One JSON object on stdout.  The driver-run suite holds the same cases
against committed digests of the same records (tests/test_gpu_fullsong.py)."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import audio_separator_amd as A  # noqa: E402
import fullsong_cases as FC  # noqa: E402
import fullsong_gpu as FG  # noqa: E402

TOL = 1e-4


def cmp(got, ref):
    g, r = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = float(np.sqrt(np.mean((g - r) ** 2)))
    rms = float(np.sqrt(np.mean(r ** 2)))
    return {"rel_rms": err / max(rms, 1e-30), "abs_rms_err": err, "max_abs_err": float(np.abs(g - r).max()), "ref_rms_in_windows": rms}


def load(name):
    path = os.path.join(FC.CACHE, name + ".npz")
    if not os.path.exists(path):
        print(f"{name}: no oracle record, computing it here", file=sys.stderr, flush=True)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "fullsong_oracle.py"), "--cases", name])
    z = np.load(path)
    return z, json.loads(str(z["meta"]))


def run_case(name):
    z, meta = load(name)
    ref = FG.record_arrays(z)                                                # {key: [..., nwin, w]}
    t0 = time.perf_counter()
    arrays, engines = FG.RUN[name](A, meta["seconds"], z)
    dt = time.perf_counter() - t0
    starts = z["starts"] if "starts" in z.files else np.zeros(1, np.int64)
    out = {"arrays": {}, "gpu_wall_s_incl_pcie_and_load": round(dt, 3)}
    for key, r in ref.items():
        g = FC.take(arrays[key], starts, r.shape[-1])
        if key.endswith("_pcm"):
            dq = np.abs(g.astype(np.int64) - r.astype(np.int64))
            out["arrays"][key] = {"pcm16_max_lsb_diff": int(dq.max()), "pcm16_frac_samples_differing": float((dq > 0).mean())}
            continue
        c = cmp(g, r)
        if r.ndim == 4 and r.shape[0] > 1:                                   # per stem too
            c["per_stem_rel_rms"] = [cmp(g[i], r[i])["rel_rms"] for i in range(r.shape[0])]
        out["arrays"][key] = c
    for e in engines:
        e.close()
    worst = max(c["rel_rms"] for c in out["arrays"].values() if "rel_rms" in c)
    out.update({"what": meta["what"], "seconds": meta["seconds"], "worst_rel_rms": worst, "pass": bool(worst < TOL), "whole_song_stats": meta["stats"],
                "cpu_oracle": {k: meta[k] for k in ("cpu_wall_s", "cpu_threads", "cpu_rtf", "host")}})
    if "scale" in z.files:
        out["final_conv_scale"] = float(z["scale"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default=",".join(FC.CASES))
    args = ap.parse_args()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    res = {"tolerance_rel_rms": TOL,
           "metric": "GPU (libasx.so, fp32) vs CPU oracle (torch / numpy fp32) on the whole workload, compared at 16 windows of 32768 samples "
                     "spread over the song (first at 0, last ending at N; the whole array for the 10-s VR case)",
           "cases": {}}
    for name in args.cases.split(","):
        try:
            r = run_case(name)
        except Exception as e:
            r = {"error": f"{type(e).__name__}: {e}"}
        res["cases"][name] = r
        print(name, json.dumps(r)[:600], file=sys.stderr, flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
