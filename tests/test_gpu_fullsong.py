"""Whole-song parity in the driver-run suite: BASELINE config 1 (UVR-MDX-NET-Inst_HQ_3 geometry, 4 minutes, 44.1 kHz stereo)
through the HIP engine -- normalise, demix (55 chunks, full-size net), * peak, secondary = mix - compensate * primary, the
writer's int16 pass -- against a committed DIGEST of the CPU oracle's output on the same seeded song and weights
(tests/golden/make_fullsong_digest.py: per window a run of 2048 consecutive samples and every 64th sample of 32768, 16 windows
from the first to the last sample of the song).  The oracle run itself is minutes of CPU and cannot be in the suite; its digest
can.  Bar: 1e-4 relative RMS (north star), int16 stream within 1 LSB."""
import json
import os

import numpy as np
import pytest

from oracle import mdx_oracle as O

pytestmark = pytest.mark.gpu


def rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


def pick(arr, z):
    """arr [2, N] -> (runs [2, nwin, run], decimated [2, nwin, width / dec]) at the digest's sample positions"""
    starts, w, run, dec, off = z["starts"], int(z["width"]), int(z["run"]), int(z["dec"]), z["run_offset"]
    runs = np.stack([arr[:, s + o:s + o + run] for s, o in zip(starts, off)], 1)
    decs = np.stack([arr[:, s:s + w:dec] for s in starts], 1)
    return runs, decs


def test_whole_song_hq3_vs_oracle_digest(golden_dir):
    import audio_separator_amd as A
    z = np.load(os.path.join(golden_dir, "fullsong_mdx_hq3_digest.npz"))
    n = int(44100 * float(z["seconds"]))
    d = O.NetDims()
    sd = O.make_convtdf_state(d, seed=0)
    for k in ("final_conv.0.weight", "final_conv.0.bias"):   # the oracle run's calibration (stem RMS ~0.1), stored with the digest
        sd[k] = sd[k] * float(z["scale"])
    p = O.MDXParams()
    eng = A.Engine(A.MDXConfig())
    eng.load_net(A.NetConfig(), A.fold_convtdf_state(sd, d.num_blocks, d.l))
    assert eng.plan(n)["n_chunks"] == 55
    mix = O.synth_mix(n, seed=0)
    primary, secondary = eng.separate(mix, 0.9, 0.0, p.compensate)
    stats = json.loads(str(z["meta"]))["stats"]
    worst = 0.0
    for nm, arr in (("primary", primary), ("secondary", secondary)):
        rows = np.ascontiguousarray(arr.T)                     # [2, N]
        assert np.isfinite(rows).all()
        runs, decs = pick(rows, z)
        e_run, e_dec = rel_rms(runs, z[nm + "_run"]), rel_rms(decs, z[nm + "_dec"])
        print(f"whole song {nm}: rel-RMS runs {e_run:.3e}, decimated {e_dec:.3e}")
        worst = max(worst, e_run, e_dec)
        # whole-array statistics of the oracle's stems
        rms = float(np.sqrt(np.mean(rows.astype(np.float64) ** 2)))
        assert abs(rms - stats[nm]["rms"]) <= 1e-4 * stats[nm]["rms"]
        assert abs(float(np.abs(rows).max()) - stats[nm]["peak"]) <= 1e-4 * stats[nm]["peak"]
        pcm, _ = eng.pcm16(arr, 0.9, 0.0)
        pr, pd = pick(np.ascontiguousarray(pcm.T), z)
        dq = max(int(np.abs(pr.astype(np.int64) - z[nm + "_pcm_run"]).max()), int(np.abs(pd.astype(np.int64) - z[nm + "_pcm_dec"]).max()))
        assert dq <= 1, dq
    eng.close()
    assert worst < 1e-4, worst
