#!/usr/bin/env python3
"""Digests of the CPU oracle's output on the WHOLE workload of every BASELINE config, for the driver-run GPU tests in
tests/test_gpu_fullsong.py.  The oracle runs themselves are minutes of CPU each (tools/fullsong_oracle.py ->
gpurun_cache/fullsong/<case>.npz: 16 windows of 32768 samples per song, the whole array for the 10-s VR clip); a digest keeps,
per window, one contiguous run of 2048 samples (anchored at the song's first / last sample for the first / last window) plus
every 64th sample of the whole window, of every array the record holds (float stems; the writer's int16 stream for the MDX case),
the whole-song statistics, and the seeds-only parameters the GPU leg needs (the MDX calibration scale).  Weights and input are
rebuilt from seeds by the test.

    python tools/fullsong_oracle.py --cases <case,...>     # only for records missing under gpurun_cache/fullsong/
    python tests/golden/make_fullsong_digest.py [case ...]  # default: every case of tools/fullsong_cases.py
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fullsong_cases as FC  # noqa: E402
import fullsong_gpu as FG  # noqa: E402

RUN, DEC = 2048, 64


def digest(name):
    src = os.path.join(FC.CACHE, name + ".npz")
    if not os.path.exists(src):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "fullsong_oracle.py"), "--cases", name])
    z = np.load(src)
    meta = json.loads(str(z["meta"]))
    arrays = FG.record_arrays(z)
    first = next(iter(arrays.values()))
    nwin, w = first.shape[-2], first.shape[-1]
    starts = z["starts"].astype(np.int64) if "starts" in z.files else np.zeros(1, np.int64)
    off = np.zeros(nwin, np.int64)
    off[-1] = w - RUN                        # the last run ends at the last sample
    out = {"starts": starts, "width": w, "run": RUN, "dec": DEC, "run_offset": off, "seconds": meta["seconds"],
           "meta": json.dumps({"case": name, "what": meta["what"], "stats": meta["stats"], "keys": sorted(arrays),
                               "source": f"tools/fullsong_oracle.py --cases {name} (oracle/*.py, torch-CPU / numpy fp32)",
                               "cpu_wall_s": meta.get("cpu_wall_s"), "cpu_threads": meta.get("cpu_threads")})}
    if "scale" in z.files:
        out["scale"] = z["scale"]
    for nm, a in arrays.items():             # [..., nwin, w]
        out[nm + "_run"] = np.stack([a[..., i, off[i]:off[i] + RUN] for i in range(nwin)], -2)
        out[nm + "_dec"] = a[..., ::DEC]
    path = os.path.join(HERE, f"fullsong_{name}_digest.npz")
    np.savez_compressed(path, **out)
    print(name, os.path.getsize(path) >> 10, "KiB", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})


if __name__ == "__main__":
    for case in (sys.argv[1:] or list(FC.CASES)):
        digest(case)
