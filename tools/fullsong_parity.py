#!/usr/bin/env python3
"""GPU leg of the whole-song parity records (one per BASELINE config): the HIP engine on the same seeded inputs as
tools/fullsong_oracle.py, compared with the oracle's stored output at the comparison windows of tools/fullsong_cases.py.

    python tools/fullsong_parity.py [--cases mdx_hq3,htdemucs,hdemucs_mmi,vr_2hp,vr_2hp_sinc,mdx23c] > profiles/r03_fullsong_parity.json

A case whose oracle record (gpurun_cache/fullsong/<case>.npz) is missing is computed on the spot with the oracle (slow: the
CPU leg of a 4-minute htdemucs song is minutes).  Reported per stem: relative RMS error (the north-star metric, bar 1e-4),
ABSOLUTE RMS error and the stem's own RMS / peak -- the synthetic nets are scaled so that stems are O(0.1), i.e. the 0.9
normalisation threshold, the 1e-6 silence threshold and the int16 quantisation run in their real range; for the MDX case the
writer's int16 stream is compared too (LSB differences).  One JSON object on stdout."""
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


def gpu_mdx(z, meta):
    from oracle import mdx_oracle as O
    n = int(FC.SR * meta["seconds"])
    d, sd = FC.mdx_state(float(z["scale"]))
    p = O.MDXParams()
    eng = A.Engine(A.MDXConfig())
    eng.load_net(A.NetConfig(), A.fold_convtdf_state(sd, d.num_blocks, d.l))
    mix = FC.synth(n, seed=0)
    t0 = time.perf_counter()
    primary, secondary = eng.separate(mix, 0.9, 0.0, p.compensate)
    dt = time.perf_counter() - t0
    starts, w = z["starts"], int(z["width"])
    out = {"stems": {}, "gpu_wall_s_incl_pcie": round(dt, 3)}
    for nm, arr in (("primary", primary), ("secondary", secondary)):
        c = cmp(FC.take(np.ascontiguousarray(arr.T), starts, w), z[nm])
        pcm, peak = eng.pcm16(arr, 0.9, 0.0)
        dq = np.abs(FC.take(np.ascontiguousarray(pcm.T), starts, w).astype(np.int64) - z[nm + "_pcm"].astype(np.int64))
        c.update({"pcm16_max_lsb_diff": int(dq.max()), "pcm16_frac_samples_differing": float((dq > 0).mean()), "peak_after_normalize": float(peak),
                  "whole_song": meta["stats"][nm]})
        out["stems"][nm] = c
    eng.close()
    return out


def gpu_demucs(z, meta, v3):
    n = int(FC.SR * meta["seconds"])
    mix = FC.synth(n, seed=0)
    eng = A.Engine(A.MDXConfig(n_fft=4096, hop_length=1024, dim_f=2048, segment_size=8))
    if v3:
        from oracle import hdemucs_oracle as H
        oc = H.HDConfig(segment=44)
        eng.load_hd(A.HDConfig(segment=44), H.make_hd_state(oc, 0))
        fn = eng.hd_demix
    else:
        from oracle import demucs_oracle as D
        oc = D.HTConfig()
        eng.load_ht(A.HTConfig(segment=FC.segment_fraction()), D.make_ht_state(oc, 0))
        fn = eng.ht_demix
    t0 = time.perf_counter()
    got = fn(mix, shifts=2, offsets=list(FC.OFFSETS), overlap=0.25, standardize=True, swap01=True)
    dt = time.perf_counter() - t0
    starts, w = z["starts"], int(z["width"])
    g = FC.take(got, starts, w)
    out = {"stems": {}, "gpu_wall_s_incl_pcie": round(dt, 3), "all": cmp(g, z["stems"])}
    for i, nm in enumerate(oc.sources if hasattr(oc, "sources") else range(g.shape[0])):
        c = cmp(g[i], z["stems"][i])
        c["whole_song"] = meta["stats"]["stems"][i]
        out["stems"][str(nm)] = c
    eng.close()
    return out


def gpu_vr(z, meta, res="polyphase"):
    n = int(FC.SR * meta["seconds"])
    wave = FC.synth(n, seed=1)
    from oracle import vr_oracle as V
    arch = 123821
    dm = A.VRDemixer({"model_params": FC.VR_MP, "primary_stem_name": "Instrumental", "torch_device": 0},
                     {"window_size": 512, "batch_size": 4, "aggression": 5, "asx_res_type": res}, state_dict=V.make_vr_state(arch, 0), nn_arch_size=arch)
    t0 = time.perf_counter()
    gp, gs = dm.separate_stems(wave)
    dt = time.perf_counter() - t0
    out = {"stems": {}, "gpu_wall_s_incl_pcie": round(dt, 3)}
    for nm, arr in (("primary", gp), ("secondary", gs)):
        c = cmp(arr, z[nm])
        c["whole_song"] = meta["stats"][nm]
        out["stems"][nm] = c
    dm.engine.close()
    return out


def gpu_mdx23c(z, meta):
    from oracle import mdxc_oracle as M
    n = int(FC.SR * meta["seconds"])
    mix = FC.synth(n, seed=2)
    cfg = M.V3Config()
    dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0}, {"overlap": 4}, state_dict=M.make_v3_state(cfg, 0))
    t0 = time.perf_counter()
    got = dm.engine.mdxc_demix(mix, 4)
    dt = time.perf_counter() - t0
    if got.ndim == 2:
        got = got[None]
    ref = z["stems"]
    if ref.ndim == 3:
        ref = ref[None]
    starts, w = z["starts"], int(z["width"])
    g = FC.take(got, starts, w)
    out = {"stems": {}, "gpu_wall_s_incl_pcie": round(dt, 3), "all": cmp(g, ref)}
    st = meta["stats"]["stems"]
    for i in range(g.shape[0]):
        c = cmp(g[i], ref[i])
        c["whole_song"] = st[i] if i < len(st) else None
        out["stems"][str(i)] = c
    dm.engine.close()
    return out


RUN = {"mdx_hq3": gpu_mdx, "htdemucs": lambda z, m: gpu_demucs(z, m, False), "hdemucs_mmi": lambda z, m: gpu_demucs(z, m, True),
       "vr_2hp": gpu_vr, "vr_2hp_sinc": lambda z, m: gpu_vr(z, m, "sinc_fastest"), "mdx23c": gpu_mdx23c}


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
            z, meta = load(name)
            r = RUN[name](z, meta)
            worst = max(c["rel_rms"] for c in r["stems"].values())
            r.update({"what": meta["what"], "seconds": meta["seconds"], "worst_rel_rms": worst, "pass": bool(worst < TOL),
                      "cpu_oracle": {k: meta[k] for k in ("cpu_wall_s", "cpu_threads", "cpu_rtf", "host")}})
            if "scale" in z.files:
                r["final_conv_scale"] = float(z["scale"])
        except Exception as e:
            r = {"error": f"{type(e).__name__}: {e}"}
        res["cases"][name] = r
        print(name, json.dumps(r)[:600], file=sys.stderr, flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
