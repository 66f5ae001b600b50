"""BS-Roformer oracle against golden vectors written by the reference BSRoformer / MDXCSeparator."""
import os

import numpy as np
import pytest

from oracle import roformer_oracle as R


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


CFG = R.RoformerConfig(dim=32, depth=2, heads=2, dim_head=64, freqs_per_bands=(2, 2, 4, 8, 17), stft_n_fft=64,
                       stft_hop_length=16, stft_win_length=64, dim_t=21, sample_rate=100, mlp_expansion_factor=2)
CFG2 = R.RoformerConfig(dim=32, depth=1, heads=2, dim_head=64, freqs_per_bands=(2, 2, 4, 8, 17), stft_n_fft=64,
                        stft_hop_length=16, stft_win_length=64, dim_t=21, sample_rate=100, num_stems=2,
                        time_transformer_depth=2, freq_transformer_depth=2, target_instrument=None)


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "roformer_small.npz"))


def test_forward(g):
    w = (0.4 * np.random.default_rng(81).standard_normal((2, 2, 320))).astype(np.float32)
    y = R.roformer_forward(w, R.make_roformer_state(CFG, 7), CFG)
    assert y.shape == g["fwd1"].shape == (2, 2, 320)
    assert rel_rms(y, g["fwd1"]) < 1e-5
    assert rel_rms(g["fwd1_flash"], g["fwd1"]) < 1e-5          # the reference's two attention paths agree
    y2 = R.roformer_forward(w, R.make_roformer_state(CFG2, 8), CFG2)
    assert y2.shape == g["fwd2"].shape == (2, 2, 2, 320)
    assert rel_rms(y2, g["fwd2"]) < 1e-5


@pytest.mark.parametrize("name,n,ov", [("n1000_ov8", 1000, 8), ("n1000_ov2", 1000, 2), ("n320_ov1", 320, 1),
                                       ("n777_ov2", 777, 2.5)])
def test_demix_single_stem(g, name, n, ov):
    mix = (0.4 * np.random.default_rng(90 + n).standard_normal((2, n))).astype(np.float32)
    out = R.roformer_demix(mix, R.make_roformer_state(CFG, 7), CFG, overlap=ov)
    assert out.shape == (2, 2, n)
    assert rel_rms(out[0], g[f"demix1_{name}_primary"]) < 1e-5
    assert rel_rms(mix - out[0], g[f"demix1_{name}_secondary"]) < 1e-5


def test_demix_two_stems(g):
    mix = (0.4 * np.random.default_rng(1090).standard_normal((2, 1000))).astype(np.float32)
    out = R.roformer_demix(mix, R.make_roformer_state(CFG2, 8), CFG2, overlap=2)
    assert rel_rms(out, g["demix2"]) < 1e-5


def test_plan_default_overlap_is_no_overlap():
    # overlap = 8 (seconds) * sample_rate exceeds the chunk -> step == chunk_size
    c = R.RoformerConfig(freqs_per_bands=R.DEFAULT_FREQS_PER_BANDS)
    cs, step, starts = R.roformer_plan(44100 * 240, c, 8)
    assert cs == step == 352800 and len(starts) == 30 and starts[-1] == 44100 * 240 - 352800


# ---- Mel-Band Roformer (mel_band_roformer.py) -------------------------------------------------------------------
GM = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "melroformer_small.npz"))


def mel_cfgs():
    c1 = R.RoformerConfig.mel_config(dim=32, depth=2, heads=2, dim_head=64, num_bands=6, stft_n_fft=64, stft_hop_length=16,
                                     stft_win_length=64, dim_t=21, sample_rate=100, mlp_expansion_factor=2)
    c2 = R.RoformerConfig.mel_config(dim=32, depth=1, heads=2, dim_head=64, num_bands=8, stft_n_fft=64, stft_hop_length=16,
                                     stft_win_length=64, dim_t=21, sample_rate=100, num_stems=2, time_transformer_depth=2,
                                     freq_transformer_depth=1, target_instrument=None, mask_estimator_depth=2)
    return c1, c2


def test_mel_layout():
    st, ct = R.mel_band_layout(44100, 2048, 60)
    assert len(st) == 60 and st[0] == 0 and st[-1] + ct[-1] == 1025
    assert all(a <= b for a, b in zip(st, st[1:])) and sum(ct) > 1025      # overlapping bands


def test_mel_forward_and_demix_golden():
    c1, c2 = mel_cfgs()
    w = (0.4 * np.random.default_rng(181).standard_normal((2, 2, 320))).astype(np.float32)
    sd1, sd2 = R.make_roformer_state(c1, 17), R.make_roformer_state(c2, 18)
    for got, want in ((R.roformer_forward(w, sd1, c1), GM["fwd1"]), (R.roformer_forward(w, sd2, c2), GM["fwd2"])):
        assert np.abs(got - want).max() / np.abs(want).max() < 2e-5
    mix = (0.4 * np.random.default_rng(2090).standard_normal((2, 1000))).astype(np.float32)
    d1 = R.roformer_demix(mix, sd1, c1, overlap=2)
    assert np.abs(d1[0] - GM["demix1_primary"]).max() / np.abs(GM["demix1_primary"]).max() < 2e-5
    d2 = R.roformer_demix(mix, sd2, c2, overlap=8)
    assert np.abs(d2 - GM["demix2"]).max() / np.abs(GM["demix2"]).max() < 2e-5
