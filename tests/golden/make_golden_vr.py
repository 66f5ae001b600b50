#!/usr/bin/env python3
"""Golden vectors for the VR path from the REFERENCE functions and classes (build container only).

`librosa` is absent: the reference's spec_utils is driven with a stand-in module exposing the three restatements
`stft`, `istft`, `resample` of oracle/vr_oracle.py (flagged there as unpinned against librosa itself; resample is
scipy.signal.resample_poly, exactly what librosa calls for res_type="polyphase").  `soundfile`, `audioread`, `six`
are empty stubs (file I/O only).  Everything else -- band logic, filters, nets -- is the reference's own code.

    python tests/golden/make_golden_vr.py
"""
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import vr_oracle as V  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_stub("librosa", stft=V.lr_stft, istft=V.lr_istft, resample=V.lr_resample)
_stub("soundfile")
_stub("audioread")
_stub("pydub", AudioSegment=object)
try:
    import six  # noqa: F401
except ImportError:
    _stub("six", PY2=False)
for pkg, path in (("audio_separator", f"{REF}/audio_separator"), ("audio_separator.separator", f"{REF}/audio_separator/separator"),
                  ("audio_separator.separator.uvr_lib_v5", f"{REF}/audio_separator/separator/uvr_lib_v5")):
    m = _stub(pkg)
    m.__path__ = [path]
    m.__spec__ = importlib.machinery.ModuleSpec(pkg, None, is_package=True)

from audio_separator.separator.uvr_lib_v5 import spec_utils  # noqa: E402
from audio_separator.separator.uvr_lib_v5.vr_network import nets  # noqa: E402

spec_utils.wav_resolution = "polyphase"      # the reference's ARM / MPS setting (spec_utils.py:33)

SMALL_CAP = [(2, 4), (2, 4), (6, 4, 1, 1, 0), (4, 4), (10, 4, 1, 1, 0), (4, 8), (8, 2, 1), (4, 2, 1), (4, 2, 1)]


class MP:   # ModelParameters without the file read
    def __init__(self, param):
        self.param = param


def ref_loading_mix(wave, mp):
    """VRSeparator.loading_mix (vr_separator.py:255-291) minus the file decode."""
    X_wave, X_spec_s = {}, {}
    bands_n = len(mp.param["band"])
    for d in range(bands_n, 0, -1):
        bp = mp.param["band"][d]
        if d == bands_n:
            X_wave[d] = wave
        else:
            X_wave[d] = sys.modules["librosa"].resample(X_wave[d + 1], orig_sr=mp.param["band"][d + 1]["sr"], target_sr=bp["sr"],
                                                        res_type=bp["res_type"])
        X_spec_s[d] = spec_utils.wave_to_spectrogram(X_wave[d], bp["hl"], bp["n_fft"], mp, band=d, is_v51_model=False)
    return spec_utils.combine_spectrograms(X_spec_s, mp, is_v51_model=False)


def ref_inference(X_spec, model, window_size, batch_size, aggressiveness, tta, post, thres, non_accom=False):
    """The reference's own VRSeparator.inference_vr (vr_separator.py:293-366), bound to a bare instance that carries exactly
    the attributes the method reads."""
    import logging
    from audio_separator.separator.architectures.vr_separator import VRSeparator
    inst = VRSeparator.__new__(VRSeparator)
    inst.logger = logging.getLogger("golden")
    inst.model_run = model
    inst.window_size = window_size
    inst.batch_size = batch_size
    inst.enable_tta = tta
    inst.enable_post_process = post
    inst.post_process_threshold = thres
    inst.primary_stem_name = "Vocals" if non_accom else "Instrumental"
    y_spec, v_spec = inst.inference_vr(X_spec, torch.device("cpu"), aggressiveness)
    return y_spec, v_spec, None


def main():
    out = {}
    mp_o = V.small_params()
    mp = MP(mp_o.param)
    rng = np.random.default_rng(3)
    n = 8000 * 3 + 137
    t = np.arange(n) / 8000.0
    wave = np.stack([0.3 * np.sin(2 * np.pi * 220 * t) + 0.1 * rng.standard_normal(n),
                     0.2 * np.sin(2 * np.pi * 330 * t + 1.0) + 0.1 * rng.standard_normal(n)]).astype(np.float32)
    out["wave"] = wave
    X_spec = ref_loading_mix(wave, mp)
    out["X_spec"] = X_spec
    # net forward (tiny capacity, arch ids exercising the 5- and 7-branch ASPP)
    for tag, arch, seed in (("hp", 123821, 5), ("sp7", 33966, 6)):
        model = nets.CascadedASPPNet(mp.param["bins"] * 2, SMALL_CAP, arch)
        sd = V.make_vr_state(arch, seed, SMALL_CAP)
        assert set(model.state_dict().keys()) == set(sd.keys()), sorted(set(model.state_dict().keys()) ^ set(sd.keys()))[:6]
        model.load_state_dict(sd)
        model.eval()
        model.offset = 16
        x = np.abs(rng.standard_normal((2, 2, mp.param["bins"] + 1, 64))).astype(np.float32)
        with torch.no_grad():
            out[f"{tag}_net_in"] = x
            out[f"{tag}_net_out"] = model.forward(torch.from_numpy(x)).numpy()
        if tag == "hp":
            aggr = {"value": 0.05, "split_bin": mp.param["band"][1]["crop_stop"], "aggr_correction": None}
            for name, tta, post in (("plain", False, False), ("tta", True, False), ("post", False, True)):
                y, v, mask = ref_inference(X_spec, model, 64, 2, aggr, tta, post, 0.2)
                out[f"inf_{name}_y"] = y.astype(np.complex64)
                out[f"inf_{name}_v"] = v.astype(np.complex64)
                if name == "plain":
                    out["wav_y"] = spec_utils.cmb_spectrogram_to_wave(y, mp, is_v51_model=False)
                    out["wav_v"] = spec_utils.cmb_spectrogram_to_wave(v, mp, is_v51_model=False)
                    # high_end_process (vr_separator.py:286-288, 368-372)
                    bp = mp.param["band"][3]
                    top = spec_utils.wave_to_spectrogram(wave, bp["hl"], bp["n_fft"], mp, band=3, is_v51_model=False)
                    h = (bp["n_fft"] // 2 - bp["crop_stop"]) + (mp.param["pre_filter_stop"] - mp.param["pre_filter_start"])
                    ihe = top[:, bp["n_fft"] // 2 - h: bp["n_fft"] // 2, :]
                    out["he_wav_y"] = spec_utils.cmb_spectrogram_to_wave(y, mp, h, spec_utils.mirroring("mirroring", y, ihe, mp), is_v51_model=False)
                    out["he_wav_v"] = spec_utils.cmb_spectrogram_to_wave(v, mp, h, spec_utils.mirroring("mirroring", v, ihe, mp), is_v51_model=False)
    # a mid-side parameter set through analysis + synthesis
    pm = dict(mp_o.param)
    pm["mid_side"] = True
    mpm = MP(pm)
    Xm = ref_loading_mix(wave, mpm)
    out["ms_X_spec"] = Xm
    out["ms_wav"] = spec_utils.cmb_spectrogram_to_wave(Xm.copy(), mpm, is_v51_model=False)
    np.savez_compressed(os.path.join(HERE, "vr_small.npz"), **out)
    main_v51(wave)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})


def main_v51(wave):
    """VR 5.1: nets_new.CascadedNet + the is_v51_model branches of spec_utils -> vr51_small.npz"""
    from audio_separator.separator.uvr_lib_v5.vr_network import nets_new
    out = {}
    mpo = V.small_params_v51()
    mp = MP(mpo.param)
    rng = np.random.default_rng(13)
    X_wave, X_spec_s = {}, {}
    bands_n = len(mp.param["band"])
    for d in range(bands_n, 0, -1):
        bp = mp.param["band"][d]
        if d == bands_n:
            X_wave[d] = wave
        else:
            X_wave[d] = sys.modules["librosa"].resample(X_wave[d + 1], orig_sr=mp.param["band"][d + 1]["sr"], target_sr=bp["sr"],
                                                        res_type=bp["res_type"])
        X_spec_s[d] = spec_utils.wave_to_spectrogram(X_wave[d], bp["hl"], bp["n_fft"], mp, band=d, is_v51_model=True)
    X_spec = spec_utils.combine_spectrograms(X_spec_s, mp, is_v51_model=True)
    out["X_spec"] = X_spec
    nout, nout_lstm = 16, 16
    model = nets_new.CascadedNet(mp.param["bins"] * 2, 51000, nout=nout, nout_lstm=nout_lstm)
    sd = V.make_vr51_state(mp.param["bins"] * 2, nout, nout_lstm, 9)
    assert set(model.state_dict().keys()) == set(sd.keys()), sorted(set(model.state_dict().keys()) ^ set(sd.keys()))[:6]
    model.load_state_dict(sd)
    model.eval()
    model.offset = 16
    x = np.abs(rng.standard_normal((2, 2, mp.param["bins"] + 1, 64))).astype(np.float32)
    with torch.no_grad():
        out["net_in"] = x
        out["net_out"] = model.forward(torch.from_numpy(x)).numpy()
    aggr = {"value": 0.05, "split_bin": mp.param["band"][1]["crop_stop"], "aggr_correction": None}
    y, v, mask = ref_inference(X_spec, model, 64, 2, aggr, False, False, 0.2)
    out["wav_y"] = spec_utils.cmb_spectrogram_to_wave(y, mp, is_v51_model=True)
    out["wav_v"] = spec_utils.cmb_spectrogram_to_wave(v, mp, is_v51_model=True)
    np.savez_compressed(os.path.join(HERE, "vr51_small.npz"), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
