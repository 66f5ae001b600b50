"""The CPU oracle (oracle/mdx_oracle.py) against golden vectors produced by the
reference itself (tests/golden/make_golden.py).  Pins the oracle."""
import os

import numpy as np
import pytest

from oracle import mdx_oracle as O


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_stft_small(golden_dir):
    g = load(golden_dir, "stft_small.npz")
    n_fft, hop, dim_f = int(g["n_fft"]), int(g["hop"]), int(g["dim_f"])
    x = (0.5 * np.random.default_rng(int(g["x_seed"])).standard_normal((2, 2, hop * 15))).astype(np.float32)
    X = O.stft_forward(x, n_fft, hop, dim_f)
    assert X.shape == g["X"].shape
    assert rel_rms(X, g["X"]) < 2e-6
    S = np.random.default_rng(int(g["s_seed"])).standard_normal(X.shape).astype(np.float32)
    y = O.stft_inverse(S, n_fft, hop)
    assert y.shape == g["y"].shape
    assert rel_rms(y, g["y"]) < 2e-6


def test_stft_hq3_geometry(golden_dir):
    g = load(golden_dir, "stft_hq3_subsample.npz")
    n_fft, hop, dim_f, seg = int(g["n_fft"]), int(g["hop"]), int(g["dim_f"]), int(g["seg"])
    C = hop * (seg - 1)
    x = (0.3 * np.random.default_rng(int(g["x_seed"])).standard_normal((1, 2, C))).astype(np.float32)
    X = O.stft_forward(x, n_fft, hop, dim_f)
    assert X.shape == (1, 4, dim_f, seg)
    Xs = X[:, :, g["fsel"]][:, :, :, g["tsel"]]
    assert rel_rms(Xs, g["X_sub"]) < 2e-6
    assert abs(np.abs(X.astype(np.float64)).sum() / float(g["X_abs_sum"]) - 1) < 1e-6
    S = np.random.default_rng(int(g["s_seed"])).standard_normal((1, 4, dim_f, seg)).astype(np.float32)
    y = O.stft_inverse(S, n_fft, hop)
    assert y.shape == (1, 2, C)
    assert rel_rms(y[:, :, g["ysel"]], g["y_sub"]) < 2e-6
    assert abs(np.abs(y.astype(np.float64)).sum() / float(g["y_abs_sum"]) - 1) < 1e-6


def test_stft_shape_contract():
    # reference tests/unit/test_stft.py:73-75 and :134-138
    x = np.random.default_rng(0).random((1, 2, 16000)).astype(np.float32)
    X = O.stft_forward(x, 2048, 512, 1025)
    assert X.shape[-2:] == (1025, 16000 // 512 + 1)
    y = O.stft_inverse(np.random.default_rng(1).random((1, 2, 1025, 32)).astype(np.float32), 2048, 512)
    assert y.shape == (1, 2, 7936)


@pytest.mark.parametrize("fname,bias,wseed", [("net_small.npz", False, 3), ("net_small_bias.npz", True, 4)])
def test_net_small(golden_dir, fname, bias, wseed):
    g = load(golden_dir, fname)
    d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4, bias=bias)
    sd = O.make_convtdf_state(d, seed=wseed)
    x = np.random.default_rng(int(g["x_seed"])).standard_normal((2, 4, 32, 16)).astype(np.float32)
    y = O.convtdf_forward(x, sd, d)
    assert y.shape == g["y"].shape
    assert rel_rms(y, g["y"]) < 1e-5


VARIANTS = {"gn": ("group", 4, False, 11), "gn_bias": ("group", 4, True, 12), "bn0": ("batch", 0, False, 13), "gn_bn0": ("group", 0, True, 14),
            "notdf": ("batch", None, False, 15)}


def variant_dims(name):
    norm, bn, bias, seed = VARIANTS[name]
    return O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=bn, bias=bias, norm=norm), seed


@pytest.mark.parametrize("name", list(VARIANTS))
def test_net_variants(golden_dir, name):
    """GroupNorm(2, c) (optimizer 'adamw'), bn == 0 and bn is None forms of the reference ConvTDFNet (tests/golden/make_golden_variants.py)"""
    g = load(golden_dir, "net_variants.npz")
    d, seed = variant_dims(name)
    x = np.random.default_rng(int(g["x_seed"])).standard_normal((2, 4, 32, 16)).astype(np.float32)
    y = O.convtdf_forward(x, O.make_convtdf_state(d, seed=seed), d)
    assert rel_rms(y, g[name]) < 1e-6


def test_groupnorm_net_through_the_chunk_loop(golden_dir):
    g = load(golden_dir, "net_variants.npz")
    d, seed = variant_dims("gn")
    mix = (0.4 * np.random.default_rng(int(g["mix_seed"])).standard_normal((2, int(g["mix_n"])))).astype(np.float32)
    p = O.MDXParams(n_fft=96, hop_length=16, dim_f=32, segment_size=16, overlap=0.25)
    y = O.demix(mix, p, O.make_model_run(O.make_convtdf_state(d, seed=seed), d))
    assert rel_rms(y, g["gn_demix"]) < 1e-5


def _small_net():
    d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4, bias=False)
    return O.make_model_run(O.make_convtdf_state(d, seed=3), d)


@pytest.mark.parametrize("name,overlap,denoise,match", [
    ("ov25", 0.25, False, False), ("ov25_denoise", 0.25, True, False), ("ov0", 0.0, False, False),
    ("ov75", 0.75, False, False), ("match", 0.25, False, True)])
def test_demix_small(golden_dir, name, overlap, denoise, match):
    g = load(golden_dir, "demix_small.npz")
    N = int(g["N"])
    mix = (0.4 * np.random.default_rng(int(g["mix_seed"])).standard_normal((2, N))).astype(np.float32)
    p = O.MDXParams(n_fft=96, hop_length=16, dim_f=32, segment_size=16, overlap=overlap, enable_denoise=denoise)
    out = O.demix(mix, p, _small_net(), is_match_mix=match)
    assert out.shape == g[name].shape == (2, N)
    assert np.isfinite(out).all()
    assert rel_rms(out, g[name]) < 1e-5


@pytest.mark.parametrize("n", [1, 143, 144, 145])
def test_demix_ragged(golden_dir, n):
    g = load(golden_dir, "demix_small.npz")
    mix = (0.4 * np.random.default_rng(100 + n).standard_normal((2, n))).astype(np.float32)
    p = O.MDXParams(n_fft=96, hop_length=16, dim_f=32, segment_size=16, overlap=0.25)
    out = O.demix(mix, p, _small_net())
    ref = g[f"ragged_n{n}"]
    assert out.shape == ref.shape == (2, n)
    assert rel_rms(out, ref) < 1e-5


def test_stems_small(golden_dir):
    g = load(golden_dir, "stems_small.npz")
    mix = (0.8 * np.random.default_rng(int(g["mix_seed"])).standard_normal((2, 2000))).astype(np.float32)
    p = O.MDXParams(n_fft=96, hop_length=16, dim_f=32, segment_size=16, overlap=0.25, compensate=float(g["compensate"]))
    primary, secondary = O.separate_stems(mix, p, _small_net(), 0.9, 0.0)
    assert np.array_equal(mix, g["mix_norm"])          # in-place normalisation, bit exact
    assert rel_rms(primary, g["primary"]) < 1e-5
    assert rel_rms(secondary, g["secondary"]) < 1e-5


def test_chunk_plan_hq3():
    # N = 10,584,000 -> 55 chunks (net pass), 42 chunks (match-mix pass)
    p = O.MDXParams()
    cs, gen, pad, L, step, starts, _ = O.chunk_plan(10_584_000, p, False)
    assert (cs, gen, pad, L, step, len(starts)) == (261120, 254976, 128064, 10_715_136, 195840, 55)
    assert len(O.chunk_plan(10_584_000, p, True)[5]) == 42


def test_net_flops_hq3():
    # 0.7589 TFLOP per chunk counted on the reference class
    assert abs(O.net_flops(O.NetDims()) / 0.7589e12 - 1) < 0.01
    # exact, against FlopCounterMode on the reference's own ConvTDFNet (tests/golden/make_flops_fixture.py)
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "flops.json")) as fh:
        want = json.load(fh)
    assert O.net_flops(O.NetDims()) == want["convtdf_hq3_g48"]
    assert O.net_flops(O.NetDims(g=8)) == want["convtdf_hq3_g8"]
