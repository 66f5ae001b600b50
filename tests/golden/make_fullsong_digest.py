#!/usr/bin/env python3
"""Digest of the CPU oracle's output on the WHOLE 4-minute HQ_3-geometry song (BASELINE config 1), for the driver-run GPU test
tests/test_gpu_fullsong.py.  The oracle run itself is ~4 minutes of CPU (tools/fullsong_oracle.py -> gpurun_cache/fullsong/
mdx_hq3.npz, 16 windows of 32768 samples); the digest committed here keeps, per window, one contiguous run of 2048 samples
(anchored at the song's first / last sample for the first / last window) plus every 64th sample of the whole window, float
stems and the writer's int16 stream, and the calibration scale of the final conv.  Everything else (weights, input) is rebuilt
from seeds by the test.

    python tools/fullsong_oracle.py --cases mdx_hq3      # only if gpurun_cache/fullsong/mdx_hq3.npz is missing
    python tests/golden/make_fullsong_digest.py
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
RUN, DEC = 2048, 64


def main():
    src = os.path.join(ROOT, "gpurun_cache", "fullsong", "mdx_hq3.npz")
    if not os.path.exists(src):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "fullsong_oracle.py"), "--cases", "mdx_hq3"])
    z = np.load(src)
    meta = json.loads(str(z["meta"]))
    starts, w = z["starts"].astype(np.int64), int(z["width"])
    nwin = len(starts)
    off = np.zeros(nwin, np.int64)
    off[-1] = w - RUN                        # the last run ends at the song's last sample
    out = {"starts": starts, "width": w, "run": RUN, "dec": DEC, "run_offset": off, "scale": z["scale"],
           "seconds": meta["seconds"], "meta": json.dumps({"stats": meta["stats"], "source": "tools/fullsong_oracle.py --cases mdx_hq3 (oracle/mdx_oracle.py, torch-CPU fp32)",
                                                           "cpu_wall_s": meta.get("cpu_wall_s"), "cpu_threads": meta.get("cpu_threads")})}
    for nm in ("primary", "secondary", "primary_pcm", "secondary_pcm"):
        a = z[nm]                            # [2, nwin, w]
        out[nm + "_run"] = np.stack([a[:, i, off[i]:off[i] + RUN] for i in range(nwin)], 1)
        out[nm + "_dec"] = a[:, :, ::DEC]
    np.savez_compressed(os.path.join(HERE, "fullsong_mdx_hq3_digest.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k != "meta"})


if __name__ == "__main__":
    main()
