"""MDXC (TFC-TDF v3) oracle against golden vectors written by the reference classes."""
import os

import numpy as np
import pytest

from oracle import mdxc_oracle as M


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


CFG2 = M.V3Config(n_fft=128, hop_length=16, dim_f=64, dim_t=16, num_subbands=2, num_scales=2, num_blocks_per_scale=2,
                  num_channels_model=8, growth=8, bottleneck_factor=4)
CFG1 = M.V3Config(n_fft=128, hop_length=16, dim_f=64, dim_t=16, num_subbands=2, num_scales=2, num_blocks_per_scale=1,
                  num_channels_model=8, growth=4, bottleneck_factor=2, target_instrument="Vocals", act="relu")


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "mdxc_small.npz"))


def test_forward(g):
    w = (0.4 * np.random.default_rng(61).standard_normal((2, 2, 240))).astype(np.float32)
    y2 = M.v3_forward(w, M.make_v3_state(CFG2, 5), CFG2)
    assert y2.shape == g["fwd2"].shape == (2, 2, 2, 240)
    assert rel_rms(y2, g["fwd2"]) < 1e-5
    y1 = M.v3_forward(w, M.make_v3_state(CFG1, 6), CFG1)
    assert y1.shape == g["fwd1"].shape == (2, 2, 240)
    assert rel_rms(y1, g["fwd1"]) < 1e-5


@pytest.mark.parametrize("name,n", [("n3000", 3000), ("n100", 100), ("n241", 241)])
def test_demix_two_stem(g, name, n):
    mix = (0.4 * np.random.default_rng(70 + n).standard_normal((2, n))).astype(np.float32)
    out = M.mdxc_demix(mix, M.make_v3_state(CFG2, 5), CFG2, overlap=4)
    assert out.shape == g[f"demix2_{name}"].shape == (2, 2, n)
    assert rel_rms(out, g[f"demix2_{name}"]) < 1e-5


def test_demix_overlap8_segment_override(g):
    mix = (0.4 * np.random.default_rng(3070).standard_normal((2, 3000))).astype(np.float32)
    out = M.mdxc_demix(mix, M.make_v3_state(CFG2, 5), CFG2, overlap=8, segment_size=12)
    assert rel_rms(out, g["demix2_ov8_seg12"]) < 1e-5


def test_demix_single_target_residual(g):
    mix = (0.4 * np.random.default_rng(3070).standard_normal((2, 3000))).astype(np.float32)
    primary = M.mdxc_demix(mix, M.make_v3_state(CFG1, 6), CFG1, overlap=2)
    assert primary.shape == (2, 3000)
    assert rel_rms(primary, g["demix1_primary"]) < 1e-5
    assert rel_rms(mix - primary, g["demix1_secondary"]) < 1e-5
