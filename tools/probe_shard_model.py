#!/usr/bin/env python3
"""Shard-time model of the N > 1 bench on the ONE GPU a gpurun box has (SURVEY.md 8c's stated fallback when no 8-GPU node can be
measured).  Measures on cuda:0, bench geometry (HQ_3 net, 4-minute song, 55 chunks):

  * one rank's share of the strong-scaling mode (`bench.py --mode chunks`): asx_demix_chunks_dev over the contiguous chunk range
    rank r would own at G = 1 / 2 / 4 / 8 (55 / 28 / 14 / 7 chunks; the largest and the smallest range of each partition), per-level
    kernel times of the 7-chunk pass beside the 55-chunk pass, and the fold each rank runs (asx_finalize_dev);
  * one rank's step of the weak-scaling mode (`--mode files`): S songs back to back (S = 1 and the 8 of BASELINE config 5);

and combines them with the bytes the collectives move and a per-link xGMI rate into predicted whole-job rates:

  strong:  T(G) = max_r compute(range_r) + halo p2p (one 2.09-MB chunk to the right neighbour) + fold + gather of [2, N / G] slabs
           (rank 0 receives G - 1 slabs, each over its own point-to-point link)
  weak:    T(G) = T_step(1 GPU); the gather of step k (G - 1 ranks x S x 84.7 MB into rank 0, one link each) overlaps step k + 1

    python tools/probe_shard_model.py [--reps 3] [--link-gbps 48] > profiles/r05_scale_model.json

This is a MODEL with measured compute terms, not a scaling measurement: RCCL has not run on more than one rank (the driver's 8-GPU
run is the measurement)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audio_separator_amd as A  # noqa: E402
from audio_separator_amd.sharding import halo_chunks, owned_samples, partition_chunks  # noqa: E402
from workload import synth as W  # noqa: E402

SR, SECONDS = 44100, 240


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--link-gbps", type=float, default=48.0, help="sustained GB/s of one xGMI link for a point-to-point copy")
    args = ap.parse_args()
    d = W.NetDims()
    eng = A.Engine(A.MDXConfig(), device=0)
    eng.load_net(A.NetConfig(), A.fold_convtdf_state(W.make_convtdf_state(d, seed=0), d.num_blocks, d.l))
    n = SR * SECONDS
    plan = eng.plan(n)
    nk, C = plan["n_chunks"], plan["chunk_size"]
    mix = torch.from_numpy(W.synth_mix(n, seed=0)).cuda()
    allc = torch.zeros((nk, 2, C), dtype=torch.float32, device="cuda")
    out = torch.empty_like(mix)
    st = torch.cuda.current_stream().cuda_stream

    def timeit(fn, reps=args.reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(reps):
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))

    def per_class(fn):
        eng.profile_enable(True)
        fn()
        p = eng.profile_read()
        eng.profile_enable(False)
        return {k: round(v["ms"], 3) for k, v in p.items() if v["launches"]}

    full_ms = timeit(lambda: eng.demix_dev(mix.data_ptr(), n, out.data_ptr(), stream=st))
    fold_ms = timeit(lambda: eng.finalize_dev(allc.data_ptr(), n, out.data_ptr(), stream=st))
    res = {"what": __doc__.split("\n\n")[0], "geometry": {"seconds": SECONDS, "n_chunks": nk, "chunk_size": C, "step": plan["step"]},
           "one_gpu": {"demix_ms": round(full_ms, 2), "rtf": round(SECONDS / (full_ms * 1e-3), 1), "fold_ms": round(fold_ms, 3)},
           "link_gbps_assumed": args.link_gbps, "strong": {}, "weak": {}}
    classes_full = per_class(lambda: eng.demix_chunks_dev(mix.data_ptr(), n, 0, nk, allc.data_ptr(), stream=st))
    per_chunk_full = None
    for G in (1, 2, 4, 8):
        ranges = partition_chunks(nk, G)
        sizes = sorted({b - a for a, b in ranges})
        rec = {"ranges": ranges, "compute_ms": {}}
        for sz in sizes:
            a, b = next((a, b) for a, b in ranges if b - a == sz)
            ms = timeit(lambda: eng.demix_chunks_dev(mix.data_ptr(), n, a, b, allc[a:b].data_ptr(), stream=st))
            rec["compute_ms"][str(sz)] = round(ms, 3)
        worst = max(rec["compute_ms"].values())
        big = max(sizes)
        if G == 1:
            per_chunk_full = worst / nk
        rec["ms_per_chunk_largest_range"] = round(worst / big, 4)
        rec["per_chunk_efficiency_vs_55"] = round(per_chunk_full / (worst / big), 4)
        own = owned_samples(plan, ranges, n)
        slab_bytes = 2 * 4 * max(j1 - j0 for j0, j1 in own)
        halo_bytes = halo_chunks(plan) * 2 * C * 4 if G > 1 else 0
        halo_ms = halo_bytes / (args.link_gbps * 1e9) * 1e3
        gather_ms = (slab_bytes / (args.link_gbps * 1e9) * 1e3) if G > 1 else 0.0     # G - 1 slabs arrive on G - 1 links in parallel
        T = worst + halo_ms + fold_ms + gather_ms
        rec.update({"slab_bytes_per_rank": slab_bytes, "halo_bytes_per_boundary": halo_bytes, "halo_ms": round(halo_ms, 4), "gather_ms": round(gather_ms, 4),
                    "fold_ms": round(fold_ms, 3), "predicted_ms": round(T, 2), "predicted_rtf": round(SECONDS / (T * 1e-3), 1)})
        if G == 8:
            a, b = ranges[3]
            small = per_class(lambda: eng.demix_chunks_dev(mix.data_ptr(), n, a, b, allc[a:b].data_ptr(), stream=st))
            rec["kernel_class_ms_7_chunks"] = small
            rec["kernel_class_ms_55_chunks_scaled_to_7"] = {k: round(v * 7 / nk, 3) for k, v in classes_full.items()}
        res["strong"][str(G)] = rec
    t1 = res["strong"]["1"]["predicted_ms"]
    for G in (2, 4, 8):
        res["strong"][str(G)]["predicted_speedup_vs_1"] = round(t1 / res["strong"][str(G)]["predicted_ms"], 2)

    # weak scaling (files mode): S songs per rank and step, the stems gather of the previous step overlaps
    stem_bytes = 2 * n * 4
    for S in (1, 8):
        mixes = [torch.from_numpy(W.synth_mix(n, seed=s)).cuda() for s in range(S)]
        outs = [torch.empty_like(mixes[0]) for _ in range(S)]

        def step():
            for m, o in zip(mixes, outs):
                eng.demix_dev(m.data_ptr(), n, o.data_ptr(), stream=st)
        ms = timeit(step, reps=max(1, args.reps - 1))
        gather_ms = S * stem_bytes / (args.link_gbps * 1e9) * 1e3
        res["weak"][f"songs_per_rank_{S}"] = {
            "step_ms_one_rank": round(ms, 2), "rtf_one_gpu": round(S * SECONDS / (ms * 1e-3), 1),
            "gather_bytes_per_rank_and_step": S * stem_bytes, "gather_ms_one_link": round(gather_ms, 3),
            "gather_hidden": bool(gather_ms < ms), "predicted_rtf": {str(G): round(G * S * SECONDS / (max(ms, gather_ms) * 1e-3), 1) for G in (1, 2, 4, 8)}}
        del mixes, outs
    eng.close()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
