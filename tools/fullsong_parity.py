#!/usr/bin/env python3
"""Full-size parity, once: the WHOLE 4-minute HQ_3-geometry song with the net (plain and enable_denoise) on the GPU against
the CPU oracle, and one 8-s chunk of the ep_317 BS-Roformer layout at its full depth 12.  (The test suite checks a 12-s
excerpt through the full-size net and the no-net pass over the full song; this closes the gap the round-1 review named.)

    python tools/fullsong_parity.py [--seconds 240] [--skip-denoise] [--skip-roformer] > profiles/r02_fullsong_parity.json

Runs on the GPU box: libasx.so for the GPU leg, oracle/ (torch-CPU, 32 threads) for the reference leg -- roughly
1.2x real time per pass, i.e. ~3.5 min per MDX pass.  One JSON object on stdout."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_separator_amd as A  # noqa: E402
from oracle import mdx_oracle as O  # noqa: E402

SR = 44100


def rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--skip-denoise", action="store_true")
    ap.add_argument("--skip-roformer", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    res = {"tolerance": 1e-4, "metric": "relative RMS of the separated stem, GPU (libasx.so, fp32) vs CPU oracle (torch fp32)"}
    d = O.NetDims()
    sd = O.make_convtdf_state(d, seed=0)
    mix = O.synth_mix(int(SR * args.seconds), seed=0)
    run = O.make_model_run(sd, d)
    for name, denoise in (("mdx_hq3_plain", False), ("mdx_hq3_denoise", True)):
        if denoise and args.skip_denoise:
            continue
        eng = A.Engine(A.MDXConfig(enable_denoise=denoise))
        eng.load_net(A.NetConfig(), A.fold_convtdf_state(sd, d.num_blocks, d.l))
        t0 = time.perf_counter()
        got = eng.demix(mix)
        tg = time.perf_counter() - t0
        eng.close()
        t0 = time.perf_counter()
        ref = O.demix(mix, O.MDXParams(enable_denoise=denoise), run)
        tc = time.perf_counter() - t0
        res[name] = {"seconds": args.seconds, "chunks": len(O.chunk_plan(mix.shape[1], O.MDXParams())[5]), "rel_rms": rel_rms(got, ref),
                     "max_abs": float(np.abs(got.astype(np.float64) - ref).max()), "gpu_wall_s_incl_pcie": round(tg, 3),
                     "cpu_wall_s": round(tc, 1), "cpu_threads": torch.get_num_threads(), "pass": bool(rel_rms(got, ref) < 1e-4)}
        print(name, res[name], file=sys.stderr, flush=True)
    if not args.skip_roformer:
        from oracle import roformer_oracle as R
        cfg = R.RoformerConfig(freqs_per_bands=R.DEFAULT_FREQS_PER_BANDS)          # ep_317 layout: dim 512, depth 12
        rsd = R.make_roformer_state(cfg, 0)
        dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0, "secondary_stem_name": "other"}, {"overlap": 8},
                           state_dict=rsd, max_batch=1)
        Cn = 441 * 800
        w = O.synth_mix(Cn, seed=3)[None]
        t0 = time.perf_counter()
        ref = R.roformer_forward(w, rsd, cfg)
        tc = time.perf_counter() - t0
        got = dm.engine.rof_forward(w)
        res["bs_roformer_ep317_depth12_chunk"] = {"chunk_seconds": Cn / SR, "params_M": round(sum(int(np.prod(v.shape)) for v in rsd.values()) / 1e6, 1),
                                                  "rel_rms": rel_rms(got, ref), "cpu_wall_s": round(tc, 1), "pass": bool(rel_rms(got, ref) < 1e-4)}
        print("roformer", res["bs_roformer_ep317_depth12_chunk"], file=sys.stderr, flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
