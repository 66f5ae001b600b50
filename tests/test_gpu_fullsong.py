"""Whole-workload parity in the driver-run suite, one test per BASELINE config: the HIP engine on the whole
seeded song against a committed DIGEST of the CPU oracle's output on the same song and weights
(tests/golden/make_fullsong_digest.py: per window a run of 2048 consecutive samples and every 64th sample of 32768, 16 windows
from the first to the last sample of the song; the whole 10-s clip for VR).  The oracle runs are minutes of CPU each and cannot
be in the suite; their digests can.

  mdx_hq3      config 1  UVR-MDX-NET-Inst_HQ_3 geometry, 4 min: normalise, demix (55 chunks, full-size net), * peak,
                         secondary = mix - compensate * primary, the writer's int16 pass (within 1 LSB)
  htdemucs     config 2  4 stems, shifts 2, 4 min
  bs_roformer  config 3  ep_317 layout at its own size (dim 512, depth 12), 4 min + 1 s: 31 chunks, Hamming fold, re-anchored tail
  vr_2hp(_sinc) config 0 10 s through the VR path under either resampler
  hdemucs_mmi, mdx23c    the other two sibling loops

Bar: 1e-4 relative RMS (north star) at the digest positions, whole-array RMS / peak of every stem within 1e-4 of the oracle's."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


def pick(arr, z):
    """arr [..., N] -> (runs [..., nwin, run], decimated [..., nwin, width / dec]) at the digest's sample positions"""
    starts, w, run, dec, off = z["starts"], int(z["width"]), int(z["run"]), int(z["dec"]), z["run_offset"]
    runs = np.stack([arr[..., s + o:s + o + run] for s, o in zip(starts, off)], -2)
    decs = np.stack([arr[..., s:s + w:dec] for s in starts], -2)
    return runs, decs


def stat_list(stats, key):
    """whole-song statistics of the oracle's arrays under `key`: a dict (one array) or a list (one per stem)"""
    st = stats.get(key)
    return st if isinstance(st, list) else [st]


@pytest.mark.parametrize("case", ["mdx_hq3", "htdemucs", "bs_roformer", "vr_2hp", "vr_2hp_sinc", "hdemucs_mmi", "mdx23c"])
def test_whole_workload_vs_oracle_digest(golden_dir, case):
    import torch
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    import audio_separator_amd as A
    import fullsong_gpu as FG
    z = np.load(os.path.join(golden_dir, f"fullsong_{case}_digest.npz"))
    meta = json.loads(str(z["meta"]))
    arrays, engines = FG.RUN[case](A, float(z["seconds"]), z)
    try:
        worst = 0.0
        for key in meta["keys"]:
            got = arrays[key]
            ref_run, ref_dec = z[key + "_run"], z[key + "_dec"]
            if len(z["starts"]) == 1:                              # one window = the whole clip (VR): the lengths must agree too
                assert got.shape[-1] == int(z["width"]), (got.shape, int(z["width"]))
            runs, decs = pick(got, z)
            assert runs.shape == ref_run.shape and decs.shape == ref_dec.shape, (key, runs.shape, ref_run.shape, decs.shape, ref_dec.shape)
            if key.endswith("_pcm"):                               # the writer's int16 stream: within 1 LSB
                dq = max(int(np.abs(runs.astype(np.int64) - ref_run).max()), int(np.abs(decs.astype(np.int64) - ref_dec).max()))
                assert dq <= 1, (key, dq)
                continue
            assert np.isfinite(got).all(), key
            e_run, e_dec = rel_rms(runs, ref_run), rel_rms(decs, ref_dec)
            print(f"whole workload {case} / {key}: rel-RMS runs {e_run:.3e}, decimated {e_dec:.3e}")
            worst = max(worst, e_run, e_dec)
            # whole-array statistics of the oracle's stems (one entry per stem, or one for the array)
            sts = stat_list(meta["stats"], key)
            rows = got.reshape((len(sts), -1)) if len(sts) > 1 else got.reshape((1, -1))
            for r, st in zip(rows, sts):
                rms, peak = float(np.sqrt(np.mean(r.astype(np.float64) ** 2))), float(np.abs(r).max())
                assert abs(rms - st["rms"]) <= 1e-4 * st["rms"] + 1e-9, (key, rms, st["rms"])
                assert abs(peak - st["peak"]) <= 1e-4 * st["peak"] + 1e-9, (key, peak, st["peak"])
        print(f"whole workload {case}: worst rel-RMS {worst:.3e}")
        assert worst < TOL, (case, worst)
    finally:
        for e in engines:
            e.close()


@pytest.mark.parametrize("case", ["mdx_hq3", "htdemucs", "bs_roformer", "vr_2hp", "vr_2hp_sinc", "hdemucs_mmi", "mdx23c"])
def test_whole_workload_is_deterministic(golden_dir, case):
    """Every whole-workload case twice, on fresh engines: every output array bit for bit the same.  An RMS bar does not see a
    kernel that returns a few hundred wrong elements of 76 M on every run, different ones each time -- the rotary epilogue of tdf3_kernel did
    that in round 5 (profiles/r05_rotary_epilogue_race.txt; profiles/r06_rotary_isa.txt: the wait in front of the packed multiply is there, the
    packed form fails with either descale, the scalar form does not) -- a bit-for-bit second run does, whichever kernel it is in."""
    import torch
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    import audio_separator_amd as A
    import fullsong_gpu as FG
    z = np.load(os.path.join(golden_dir, f"fullsong_{case}_digest.npz"))
    seconds = float(z["seconds"])
    if case in ("htdemucs", "hdemucs_mmi", "bs_roformer"):
        seconds = min(seconds, 61.0)                   # the same kernels on every chunk: a minute of them is enough for a determinism check
    runs = []
    for _ in range(2):
        arrays, engines = FG.RUN[case](A, seconds, z)
        runs.append({k: np.array(v, copy=True) for k, v in arrays.items()})
        for e in engines:
            e.close()
    assert runs[0].keys() == runs[1].keys()
    for k in runs[0]:
        a, b = runs[0][k], runs[1][k]
        same = np.array_equal(a, b) if a.dtype.kind != "f" else np.array_equal(a.view(np.uint32 if a.dtype == np.float32 else np.uint64), b.view(np.uint32 if b.dtype == np.float32 else np.uint64))
        assert same, (case, k, int(np.sum(a != b)), "elements differ between two runs")
