#!/usr/bin/env python3
"""CPU leg of the whole-song parity records: runs the oracle (oracle/*.py, the pinned restatement of the reference) on a
whole BASELINE workload and stores its output at the comparison windows of tools/fullsong_cases.py.

    python tools/fullsong_oracle.py --cases mdx_hq3,htdemucs,hdemucs_mmi,vr_2hp,vr_2hp_sinc,mdx23c [--threads 8]

Writes gpurun_cache/fullsong/<case>.npz (git-ignored; it travels to the GPU box with the snapshot, where
tools/fullsong_parity.py runs the HIP engine on the same seeded inputs and compares).  The CPU work is minutes per case
(4-minute songs), which is why it is not done on the GPU box's clock."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fullsong_cases as FC  # noqa: E402


def stats(x):
    x = np.asarray(x, np.float64)
    return {"rms": float(np.sqrt(np.mean(x ** 2))), "peak": float(np.abs(x).max())}


def pcm16(stem_rows, max_peak=0.9, min_peak=0.0):
    """write_audio_pydub's arithmetic (common_separator.py:309-337) on a [N, 2] stem"""
    a = np.array(stem_rows, np.float32, copy=True)
    peak = np.abs(a).max()
    if peak > max_peak:
        a *= max_peak / peak
    elif peak < min_peak:
        a *= min_peak / peak
    return (a * 32767).astype(np.int16)


def run_mdx(seconds):
    from oracle import mdx_oracle as O
    n = int(FC.SR * seconds)
    d, sd = FC.mdx_state()
    p = O.MDXParams()
    # calibration: one chunk's worth of audio through the unscaled net -> scale of the final conv for stem RMS 0.1
    cal = FC.synth(p.hop_length * (p.segment_size - 1), seed=5)
    y = O.demix(cal, p, O.make_model_run(sd, d))
    scale = float(0.1 / np.sqrt(np.mean(y.astype(np.float64) ** 2)))
    d, sd = FC.mdx_state(scale)
    mix = FC.synth(n, seed=0)
    primary, secondary = O.separate_stems(mix, p, O.make_model_run(sd, d), 0.9, 0.0)
    starts, w = FC.windows(n)
    out = {"scale": scale, "starts": starts, "width": w,
           "primary": FC.take(np.ascontiguousarray(primary.T), starts, w), "secondary": FC.take(np.ascontiguousarray(secondary.T), starts, w),
           "primary_pcm": FC.take(np.ascontiguousarray(pcm16(primary).T), starts, w),
           "secondary_pcm": FC.take(np.ascontiguousarray(pcm16(secondary).T), starts, w)}
    return out, {"primary": stats(primary), "secondary": stats(secondary), "mix_after_normalize": stats(mix)}


def run_demucs(seconds, v3):
    n = int(FC.SR * seconds)
    mix = FC.synth(n, seed=0)
    if v3:
        from oracle import hdemucs_oracle as H
        oc = H.HDConfig(segment=44)
        sd = H.make_hd_state(oc, 0)
        src = H.demix_hdemucs(mix, sd, oc, shifts=2, overlap=0.25, offsets=list(FC.OFFSETS))
    else:
        from oracle import demucs_oracle as D
        oc = D.HTConfig()
        sd = D.make_ht_state(oc, 0)
        src = D.demix_demucs(mix, sd, oc, shifts=2, overlap=0.25, offsets=list(FC.OFFSETS))
    starts, w = FC.windows(n)
    return {"starts": starts, "width": w, "stems": FC.take(src, starts, w)}, {"stems": [stats(s) for s in src], "mix": stats(mix)}


def run_vr(seconds, res="polyphase"):
    from oracle import vr_oracle as V
    n = int(FC.SR * seconds)
    wave = FC.synth(n, seed=1)
    arch = 123821
    sd = V.make_vr_state(arch, 0)
    p, s = V.vr_separate(wave, sd, arch, V.ModelParams(FC.VR_MP), window_size=512, batch_size=2, aggression=5, wav_resolution=res)
    return {"primary": np.asarray(p, np.float32), "secondary": np.asarray(s, np.float32)}, {"primary": stats(p), "secondary": stats(s), "mix": stats(wave)}


def run_mdx23c(seconds):
    from oracle import mdxc_oracle as M
    n = int(FC.SR * seconds)
    mix = FC.synth(n, seed=2)
    cfg = M.V3Config()
    sd = M.make_v3_state(cfg, 0)
    out = M.mdxc_demix(mix, sd, cfg, overlap=4)
    starts, w = FC.windows(n)
    return {"starts": starts, "width": w, "stems": FC.take(out, starts, w)}, {"stems": [stats(s) for s in out], "mix": stats(mix)}


def run_roformer(seconds):
    from oracle import roformer_oracle as R
    n = int(FC.SR * seconds)
    mix = FC.synth(n, seed=3)
    cfg = FC.roformer_config()
    sd = R.make_roformer_state(cfg, 0)
    out = R.roformer_demix(mix, sd, cfg, overlap=8)[:1]      # [len(instruments), 2, N] with identical rows (single target): keep one
    starts, w = FC.windows(n)
    return {"starts": starts, "width": w, "stems": FC.take(out, starts, w)}, {"stems": [stats(s) for s in out], "mix": stats(mix)}


RUN = {"mdx_hq3": run_mdx, "htdemucs": lambda s: run_demucs(s, False), "hdemucs_mmi": lambda s: run_demucs(s, True), "vr_2hp": run_vr, "vr_2hp_sinc": lambda s: run_vr(s, "sinc_fastest"),
       "mdx23c": run_mdx23c, "bs_roformer": run_roformer}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default=",".join(FC.CASES))
    ap.add_argument("--threads", type=int, default=min(os.cpu_count() or 1, 32))
    ap.add_argument("--force", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    os.makedirs(FC.CACHE, exist_ok=True)
    for name in args.cases.split(","):
        path = os.path.join(FC.CACHE, name + ".npz")
        if os.path.exists(path) and not args.force:
            print(name, "cached", file=sys.stderr)
            continue
        seconds, what = FC.CASES[name]
        t0 = time.perf_counter()
        arrays, st = RUN[name](seconds)
        dt = time.perf_counter() - t0
        meta = {"case": name, "what": what, "seconds": seconds, "cpu_wall_s": round(dt, 1), "cpu_threads": args.threads,
                "cpu_rtf": round(seconds / dt, 3), "host": os.uname().nodename, "stats": st}
        np.savez(path, meta=json.dumps(meta), **arrays)
        print(name, json.dumps(meta), file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
