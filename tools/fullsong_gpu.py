"""GPU legs of the whole-workload parity cases (tools/fullsong_cases.py): the HIP engine on the same seeded weights and
input as the CPU leg (tools/fullsong_oracle.py), returning the FULL output arrays, sample axis last.

Shared by tools/fullsong_parity.py (compares at the oracle record's sample windows; builder-run, needs the 17-MB records under
gpurun_cache/) and tests/test_gpu_fullsong.py (compares at the positions of the committed digests under tests/golden/;
driver-run).  Every function takes the record / digest `z` (for the seeds it may carry, e.g. the MDX calibration scale) and
returns (arrays, extras): arrays = {key: float32 [..., N]} with the keys the oracle leg stored, extras = engine objects the
caller may still want (the MDX engine for the int16 pass) -- close() them when done."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import fullsong_cases as FC  # noqa: E402


def gpu_mdx(A, seconds, z):
    from oracle import mdx_oracle as O
    n = int(FC.SR * seconds)
    d, sd = FC.mdx_state(float(z["scale"]))
    p = O.MDXParams()
    eng = A.Engine(A.MDXConfig())
    eng.load_net(A.NetConfig(), A.fold_convtdf_state(sd, d.num_blocks, d.l))
    assert eng.plan(n)["n_chunks"] == 55
    mix = FC.synth(n, seed=0)
    primary, secondary = eng.separate(mix, 0.9, 0.0, p.compensate)          # [N, 2] each
    out = {"primary": np.ascontiguousarray(primary.T), "secondary": np.ascontiguousarray(secondary.T)}
    for nm, arr in (("primary", primary), ("secondary", secondary)):
        pcm, _peak = eng.pcm16(arr, 0.9, 0.0)
        out[nm + "_pcm"] = np.ascontiguousarray(pcm.T)
    return out, [eng]


def gpu_demucs(A, seconds, z, v3):
    n = int(FC.SR * seconds)
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
    got = fn(mix, shifts=2, offsets=list(FC.OFFSETS), overlap=0.25, standardize=True, swap01=True)   # [S, 2, N]
    return {"stems": got}, [eng]


def gpu_vr(A, seconds, z, res="polyphase"):
    from oracle import vr_oracle as V
    n = int(FC.SR * seconds)
    wave = FC.synth(n, seed=1)
    arch = 123821
    dm = A.VRDemixer({"model_params": FC.VR_MP, "primary_stem_name": "Instrumental", "torch_device": 0},
                     {"window_size": 512, "batch_size": 4, "aggression": 5, "asx_res_type": res}, state_dict=V.make_vr_state(arch, 0), nn_arch_size=arch)
    gp, gs = dm.separate_stems(wave)                                         # [n', 2] each
    return {"primary": np.ascontiguousarray(gp.T), "secondary": np.ascontiguousarray(gs.T)}, [dm.engine]


def gpu_mdx23c(A, seconds, z):
    from oracle import mdxc_oracle as M
    n = int(FC.SR * seconds)
    mix = FC.synth(n, seed=2)
    cfg = M.V3Config()
    dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0}, {"overlap": 4}, state_dict=M.make_v3_state(cfg, 0))
    got = dm.engine.mdxc_demix(mix, 4)
    if got.ndim == 2:
        got = got[None]
    return {"stems": got}, [dm.engine]


def gpu_roformer(A, seconds, z):
    from oracle import roformer_oracle as R
    n = int(FC.SR * seconds)
    mix = FC.synth(n, seed=3)
    cfg = FC.roformer_config()
    dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0, "secondary_stem_name": "other"}, {"overlap": 8},
                       state_dict=R.make_roformer_state(cfg, 0))
    out = dm.demix(mix)                                                      # {"vocals": [2, N], "other": mix - vocals}
    return {"stems": np.ascontiguousarray(out["vocals"])[None]}, [dm.engine]


RUN = {"mdx_hq3": gpu_mdx, "htdemucs": lambda A, s, z: gpu_demucs(A, s, z, False), "hdemucs_mmi": lambda A, s, z: gpu_demucs(A, s, z, True),
       "vr_2hp": gpu_vr, "vr_2hp_sinc": lambda A, s, z: gpu_vr(A, s, z, "sinc_fastest"), "mdx23c": gpu_mdx23c, "bs_roformer": gpu_roformer}


def record_arrays(z):
    """The comparable arrays of an oracle record, sample windows last: {key: [..., nwin, width]}.  VR records hold whole [n, 2]
    arrays: one window covering everything, channels first."""
    skip = {"meta", "starts", "width", "scale"}
    out = {}
    for k in z.files:
        if k in skip:
            continue
        a = z[k]
        if "starts" not in z.files:                    # whole arrays [n, 2] (VR)
            a = np.ascontiguousarray(a.T)[..., None, :]
        elif k == "stems" and a.ndim == 3:             # single-stem nets stored without the stem axis
            a = a[None]
        out[k] = a
    return out
