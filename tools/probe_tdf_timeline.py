#!/usr/bin/env python3
"""Ground truth for the row-GEMM schedule: run a 60-s HQ_3 song with the ASX_TDF2_ABL=16 build (s_memtime probes in the first
4096 workgroups of the K = 384 launches) and print, per CU, when each workgroup started, got its first stage, left the K loop
and finished its epilogue -- i.e. whether the two co-resident workgroups' epilogues overlap with each other or with the other's
MFMA loop.  Measurement aid (GPU box):  ASX_TDF2=1 ASX_TDF2_ABL=16 python tools/probe_tdf_timeline.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import audio_separator_amd as A  # noqa: E402
from audio_separator_amd.engine import load_library  # noqa: E402
from oracle import mdx_oracle as O  # noqa: E402


def main():
    d = O.NetDims()
    sd = O.make_convtdf_state(d, seed=0)
    eng = A.Engine(A.MDXConfig())
    eng.load_net(A.NetConfig(), A.fold_convtdf_state(sd, d.num_blocks, d.l))
    n = 44100 * 60
    mix = torch.from_numpy(O.synth_mix(n, seed=0)).cuda()
    out = torch.empty_like(mix)
    for _ in range(2):
        eng.demix_dev(mix.data_ptr(), n, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lib = load_library()
    buf = (C.c_uint64 * (4096 * 8))()
    lib.asx_debug_trace.argtypes = [C.POINTER(C.c_uint64), C.c_int64]
    assert lib.asx_debug_trace(buf, 4096 * 8) == 0
    t = np.frombuffer(buf, np.uint64).reshape(4096, 8).astype(np.int64)
    t = t[t[:, 0] > 0]
    base = t[:, 0].min()
    hw, xcc = t[:, 4], t[:, 5]
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 0x7
    key = (xcc & 0xF) * 4096 + se * 256 + sh * 16 + cu
    print("workgroups recorded", len(t), "grid", int(t[0, 6]), "distinct CUs", len(np.unique(key)))
    dur = t[:, 3] - t[:, 0]
    print("per-WG cycles: start->first data %.0f | K loop %.0f | epilogue %.0f | total %.0f (medians)" % (
        np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1]), np.median(t[:, 3] - t[:, 2]), np.median(dur)))
    shown = 0
    for k in np.unique(key):
        rows = t[key == k]
        rows = rows[np.argsort(rows[:, 0])]
        if len(rows) < 6:
            continue
        print(f"CU key {k:#x}: {len(rows)} workgroups")
        for r in rows[:10]:
            print("   start %9d  data +%6d  loop_end +%7d  epi_end +%7d   (wave %d simd %d)" % (
                r[0] - base, r[1] - r[0], r[2] - r[0], r[3] - r[0], r[4] & 0xF, (r[4] >> 4) & 3))
        # overlap statistics on this CU: fraction of each epilogue interval during which another WG of the CU is in its K loop
        ov = []
        for i, r in enumerate(rows):
            e0, e1 = r[2], r[3]
            cover = 0
            for j, q in enumerate(rows):
                if i != j:
                    cover += max(0, min(e1, q[2]) - max(e0, q[1]))
            ov.append(cover / max(1, e1 - e0))
        print("   mean fraction of an epilogue covered by a neighbour's K loop: %.2f" % float(np.mean(ov)))
        shown += 1
        if shown >= 3:
            break


if __name__ == "__main__":
    main()
