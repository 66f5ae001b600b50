"""Oracle pin for Demucs v3 (HDemucs): oracle/hdemucs_oracle.py against vectors written by the reference HDemucs class."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from oracle.demucs_oracle import apply_model  # noqa: E402
from oracle.hdemucs_oracle import HDConfig, hd_forward, make_hd_state  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "hdemucs_small.npz")


def small_cfg():
    return HDConfig(channels=8, nfft=1024, depth=5, norm_starts=3, dconv_attn=3, dconv_lstm=3, samplerate=8000, segment=2)


def test_layer_plan_released_model():
    # hdemucs_mmi: 4 strided frequency layers, the last_freq layer (kernel 8 over the remaining 8 rows), one time-only layer
    Ls = HDConfig().layers()
    assert [L["freqs_in"] for L in Ls] == [2048, 512, 128, 32, 8, 1]
    assert [L["last_freq"] for L in Ls] == [False, False, False, False, True, False]
    assert [L["chout_z"] for L in Ls] == [48, 96, 192, 384, 768, 1536]
    assert [L["tenc"] for L in Ls] == [True] * 5 + [False]
    assert [L["lstm"] for L in Ls] == [False] * 4 + [True] * 2


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f"])
def test_forward_matches_reference(tag):
    g = np.load(GOLD)
    cfg = small_cfg()
    sd = make_hd_state(cfg, 21)
    y = hd_forward(g[f"x_{tag}"], sd, cfg)
    ref = g[f"y_{tag}"]
    assert y.shape == ref.shape
    err = np.abs(y - ref).max() / np.abs(ref).max()
    assert err < 2e-5, err


@pytest.mark.parametrize("mode", ["split", "shift"])
def test_apply_model_matches_reference(mode):
    # chunks run unpadded at their own length (HDemucs has no valid_length, apply.py:251-256)
    g = np.load(GOLD)
    cfg = small_cfg()
    sd = make_hd_state(cfg, 21)
    fn = lambda x: hd_forward(x.numpy() if hasattr(x, "numpy") else x, sd, cfg)  # noqa: E731
    if mode == "split":
        y = apply_model(fn, g["mix"], cfg, shifts=0, split=True, overlap=0.25)
    else:
        y = apply_model(fn, g["mix"], cfg, shifts=2, split=True, overlap=0.25, offsets=[int(o) for o in g["offsets"]])
    ref = g[mode]
    err = np.abs(np.asarray(y) - ref).max() / np.abs(ref).max()
    assert err < 2e-5, err


def test_forward_wide_hidden_matches_reference():
    g = np.load(GOLD)
    cfg = HDConfig(channels=8, nfft=1024, depth=5, norm_starts=3, dconv_attn=3, dconv_lstm=3, dconv_comp=1, samplerate=8000, segment=2)
    y = hd_forward(g["x_w"], make_hd_state(cfg, 22), cfg)
    err = np.abs(y - g["y_w"]).max() / np.abs(g["y_w"]).max()
    assert err < 2e-5, err
