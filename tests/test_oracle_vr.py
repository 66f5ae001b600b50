"""Oracle (oracle/vr_oracle.py) pinned on vectors written by the reference's spec_utils functions and nets.py classes
(tests/golden/make_golden_vr.py); its librosa restatements cross-checked against scipy.signal."""
import os
import sys

import numpy as np
import pytest
import scipy.signal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vr_oracle as V  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "vr_small.npz"))
SMALL_CAP = [(2, 4), (2, 4), (6, 4, 1, 1, 0), (4, 4), (10, 4, 1, 1, 0), (4, 8), (8, 2, 1), (4, 2, 1), (4, 2, 1)]


def close(a, b, tol=2e-5):
    err = np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-30)
    assert err < tol, err


def test_stft_istft_vs_scipy():
    """librosa.stft == scipy.signal.stft up to scipy's window-sum scaling; istft inverts it (n_fft 96 = 3 * 2^5)"""
    rng = np.random.default_rng(0)
    y = rng.standard_normal(4000).astype(np.float32)
    for n_fft, hop in ((96, 64), (128, 16), (64, 32)):
        S = V.lr_stft(y, n_fft, hop)
        assert S.dtype == np.complex64 and S.shape == (n_fft // 2 + 1, 1 + len(y) // hop)
        w = V.hann_periodic(n_fft)
        _, _, Z = scipy.signal.stft(y.astype(np.float64), window=w, nperseg=n_fft, noverlap=n_fft - hop, boundary="zeros",
                                    padded=False, return_onesided=True)
        close(S, Z[:, : S.shape[1]] * w.sum(), 1e-5)
        if n_fft // hop >= 2 and n_fft % hop == 0:     # NOLA holds: perfect reconstruction inside the signal
            x = V.lr_istft(S.astype(np.complex128), hop)
            assert x.shape == (hop * (S.shape[1] - 1),)
            m = min(len(x), len(y))
            close(x[n_fft:m - n_fft], y[n_fft:m - n_fft], 1e-5)


def test_resample_lengths():
    y = np.random.default_rng(1).standard_normal((2, 1001)).astype(np.float32)
    assert V.lr_resample(y, 8000, 4000).shape == (2, 501) and V.lr_resample(y, 8000, 4000).dtype == np.float32
    assert V.lr_resample(y, 4000, 8000).shape == (2, 2002)
    assert V.lr_resample(y, 4000, 4000) is y


def test_analysis_golden():
    mp = V.small_params()
    close(V.loading_mix(G["wave"], mp), G["X_spec"], 1e-6)
    pm = dict(mp.param)
    pm["mid_side"] = True
    close(V.loading_mix(G["wave"], V.ModelParams(pm)), G["ms_X_spec"], 1e-6)
    close(V.cmb_spectrogram_to_wave(G["ms_X_spec"].copy(), V.ModelParams(pm)), G["ms_wav"], 1e-6)


@pytest.mark.parametrize("tag,arch,seed", [("hp", 123821, 5), ("sp7", 33966, 6)])
def test_net_golden(tag, arch, seed):
    sd = V.make_vr_state(arch, seed, SMALL_CAP)
    close(V.cascaded_forward(G[f"{tag}_net_in"], sd, arch, 192), G[f"{tag}_net_out"])


@pytest.mark.parametrize("name,tta,post", [("plain", False, False), ("tta", True, False), ("post", False, True)])
def test_inference_golden(name, tta, post):
    mp = V.small_params()
    sd = V.make_vr_state(123821, 5, SMALL_CAP)
    aggr = {"value": 0.05, "split_bin": mp.param["band"][1]["crop_stop"], "aggr_correction": None}
    y, v = V.inference_vr(G["X_spec"], lambda x: V.predict_mask(x, sd, 123821, 192, 16), 64, 16, 2, aggr, False, tta, post, 0.2)
    close(y, G[f"inf_{name}_y"])
    close(v, G[f"inf_{name}_v"])
    if name == "plain":
        close(V.cmb_spectrogram_to_wave(y, mp), G["wav_y"], 1e-5)
        close(V.cmb_spectrogram_to_wave(v, mp), G["wav_v"], 1e-5)


def test_separate_end_to_end():
    mp = V.small_params()
    sd = V.make_vr_state(123821, 5, SMALL_CAP)
    p, s = V.vr_separate(G["wave"], sd, 123821, mp, window_size=64, batch_size=2, aggression=5, offset=16)
    close(p.T, G["wav_y"], 1e-5)
    close(s.T, G["wav_v"], 1e-5)


# ---- VR 5.1 (nets_new.CascadedNet, is_v51_model branches) --------------------------------------------------------------
G51 = np.load(os.path.join(ROOT, "tests", "golden", "vr51_small.npz"))


def test_v51_analysis_and_net_golden():
    mp = V.small_params_v51()
    close(V.loading_mix_v51(G["wave"], mp), G51["X_spec"], 1e-6)
    sd = V.make_vr51_state(192, 16, 16, 9)
    close(V.cascaded51_forward(G51["net_in"], sd, 192), G51["net_out"])


def test_v51_separate_golden():
    mp = V.small_params_v51()
    sd = V.make_vr51_state(192, 16, 16, 9)
    p, s = V.vr_separate_v51(G["wave"], sd, mp, window_size=64, batch_size=2, aggression=5, offset=16)
    close(p.T, G51["wav_y"], 1e-5)
    close(s.T, G51["wav_v"], 1e-5)


def test_high_end_process_golden():
    mp = V.small_params()
    sd = V.make_vr_state(123821, 5, SMALL_CAP)
    p, s = V.vr_separate(G["wave"], sd, 123821, mp, window_size=64, batch_size=2, aggression=5, offset=16, high_end_process=True)
    close(p.T, G["he_wav_y"], 1e-5)
    close(s.T, G["he_wav_v"], 1e-5)
    assert np.abs(G["he_wav_y"] - G["wav_y"]).max() > 1e-4     # the option changes the result
