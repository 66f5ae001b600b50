"""Oracle (oracle/demucs_oracle.py) pinned on vectors written by the reference HTDemucs / apply_model
(tests/golden/make_golden_demucs.py)."""
import os
import sys
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import demucs_oracle as do  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "demucs_small.npz"))


def cfg_a():
    return do.HTConfig(channels=16, nfft=1024, depth=3, bottom_channels=128, t_layers=3, t_heads=2,
                       samplerate=8000, segment=Fraction(1, 1))


def cfg_b():
    return do.HTConfig(channels=24, nfft=1024, depth=3, bottom_channels=0, t_layers=2, t_heads=2,
                       samplerate=8000, segment=Fraction(1, 1))


def close(a, b, tol=2e-5):
    scale = np.abs(b).max()
    err = np.abs(a - b).max() / scale
    assert err < tol, err


@pytest.mark.parametrize("tag,cfg,seed", [("a", cfg_a(), 11), ("b", cfg_b(), 12)])
def test_forward(tag, cfg, seed):
    sd = do.make_ht_state(cfg, seed)
    close(do.ht_forward(G[f"{tag}_fwd_in"], sd, cfg), G[f"{tag}_fwd_out"])
    tl = cfg.training_length
    close(do.ht_forward(G[f"{tag}_fwd_in"][:1, :, : tl - 777], sd, cfg), G[f"{tag}_short_out"])


def test_apply_model():
    cfg = cfg_a()
    sd = do.make_ht_state(cfg, 11)
    fn = lambda x: do.ht_forward(x.numpy(), sd, cfg)  # noqa: E731
    mix = G["a_mix"]
    close(do.apply_model(fn, mix, cfg, shifts=0, split=True).numpy(), G["a_split"])
    close(do.apply_model(fn, mix, cfg, shifts=2, split=True, offsets=[int(o) for o in G["a_offsets"]]).numpy(), G["a_shift"])
    tl = cfg.training_length
    close(do.apply_model(fn, mix[..., : tl - 100], cfg, shifts=0, split=False).numpy(), G["a_nosplit"])
