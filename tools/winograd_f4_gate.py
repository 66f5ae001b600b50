#!/usr/bin/env python3
"""CPU-only feasibility gate for Winograd F(4x4, 3x3) on the TFC 3x3 convs.

The shipped kernel is F(2x2, 3x3) (csrc/kernels_wino.h): 16 transform-domain multiply-adds per 2x2 output tile instead of 36,
whole-song deviation from the CPU oracle 4.1e-6.  F(4x4, 3x3) needs 36 per 4x4 tile (4x fewer than direct, 1.78x fewer than
F(2x2)) but its transforms carry constants up to 8 / 24 and lose 2-3 bits per layer in fp32.  This script emulates both in
torch fp32 exactly where the kernel would round (U = G g G^T in fp64 -> fp32 once; V = B^T d B, the channel sums and
Y = A^T m A in fp32), runs the full HQ_3 net (33 3x3 convs, every other layer untouched) on a 12-s excerpt through the
oracle's demix loop with the 3x3 convs swapped, and reports the relative RMS deviation of the separated waveform from the
plain oracle (direct fp32 convolution).  Gate: < 3e-5 (a third of the 1e-4 bar, leaving room for the other layers' error).

    python tools/winograd_f4_gate.py [--seconds 12] [--threads 8]  > profiles/r04_winograd_f4_gate.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mdx_oracle as O  # noqa: E402

MATS = {
    2: dict(BT=[[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]],
            G=[[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]],
            AT=[[1, 1, 1, 0], [0, 1, -1, -1]]),
    4: dict(BT=[[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
            G=[[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
            AT=[[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]),
}


def wino_conv3x3(x, w, b, m, kchunk=4):
    """conv2d(x, w, b, padding=1) as Winograd F(m x m, 3 x 3), fp32 arithmetic; channel sums in k-chunks of 4 accumulated
    in order (the MFMA k-step), everything else as einsum in fp32."""
    BT = torch.tensor(MATS[m]["BT"], dtype=torch.float32)
    AT = torch.tensor(MATS[m]["AT"], dtype=torch.float32)
    G = torch.tensor(MATS[m]["G"], dtype=torch.float64)
    a = m + 2
    Bn, C, H, W = x.shape
    Co = w.shape[0]
    U = torch.einsum("ai,ocij,bj->abco", G, w.double(), G).float()           # [a, a, C, Co], rounded once
    th, tw = -(-H // m), -(-W // m)
    xp = F.pad(x, (1, 1 + tw * m - W, 1, 1 + th * m - H))
    out = torch.empty(Bn, Co, th * m, tw * m)
    rows = max(1, 64 // m)                                                     # tile rows per slab (memory)
    for r0 in range(0, th, rows):
        r1 = min(th, r0 + rows)
        slab = xp[:, :, r0 * m:r1 * m + 2, :]
        d = slab.unfold(2, a, m).unfold(3, a, m)                               # [B, C, nh, nw, a, a]
        V = torch.einsum("ai,bcnwij,ej->bnwaec", BT, d, BT)                   # [B, nh, nw, a, a, C]
        M = None
        for c0 in range(0, C, kchunk):                                         # ordered fp32 accumulation over channel quads
            t = torch.einsum("bnwaec,aeco->bnwaeo", V[..., c0:c0 + kchunk], U[:, :, c0:c0 + kchunk, :])
            M = t if M is None else M + t
        Y = torch.einsum("pa,bnwaeo,qe->bonpwq", AT, M, AT)                   # [B, Co, nh, m, nw, m]
        out[:, :, r0 * m:r1 * m, :] = Y.reshape(Bn, Co, (r1 - r0) * m, tw * m)
    out = out[:, :, :H, :W]
    return out + b.view(1, -1, 1, 1) if b is not None else out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--g", type=int, default=48)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    d = O.NetDims(g=args.g)
    sd = O.make_convtdf_state(d, seed=0)
    p = O.MDXParams()
    mix = O.synth_mix(int(44100 * args.seconds), seed=0)
    t0 = time.time()
    ref = O.demix(mix, p, O.make_model_run(sd, d))
    t_ref = time.time() - t0
    res = {"what": "whole HQ_3 net (33 TFC 3x3 convs swapped, everything else the oracle's fp32 path) on a synthetic excerpt, separated "
                   "waveform vs the plain oracle, relative RMS", "seconds": args.seconds, "g": args.g,
           "chunks": len(O.chunk_plan(mix.shape[1], p)[5]), "gate": 3e-5, "oracle_wall_s": round(t_ref, 1)}
    real = F.conv2d
    for m in (2, 4):
        stats = {"n": 0, "worst_layer_rel": 0.0}

        def patched(x, w, b=None, stride=1, padding=0, *a, **k):
            if w.shape[-1] == 3 and w.shape[-2] == 3 and stride == 1 and padding == 1:
                y = wino_conv3x3(x, w, b, m)
                if stats["n"] < 33:                      # per-layer deviation on the first chunk (same input, both algorithms)
                    yd = real(x, w, b, padding=1)
                    e = float(((y - yd).double().pow(2).mean() / yd.double().pow(2).mean()).sqrt())
                    stats["worst_layer_rel"] = max(stats["worst_layer_rel"], e)
                    stats.setdefault("layer_rel", []).append(float(f"{e:.3e}"))
                stats["n"] += 1
                return y
            return real(x, w, b, stride, padding, *a, **k)
        O.F.conv2d = patched
        try:
            t0 = time.time()
            got = O.demix(mix, p, O.make_model_run(sd, d))
            dt = time.time() - t0
        finally:
            O.F.conv2d = real
        e = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)) / np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
        res[f"F({m}x{m},3x3)"] = {"whole_net_rel_rms_vs_oracle": float(f"{e:.3e}"), "pass": bool(e < 3e-5), "convs_swapped": stats["n"],
                                  "worst_single_layer_rel_rms_vs_direct": float(f"{stats['worst_layer_rel']:.3e}"),
                                  "layer_rel_first_chunk": stats.get("layer_rel"), "wall_s": round(dt, 1)}
        print(f"F({m}x{m}): whole-net rel-RMS {e:.3e}  ({dt:.0f} s)", file=sys.stderr, flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
