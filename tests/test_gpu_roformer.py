"""GPU parity of the BS-Roformer path against golden vectors written by the reference BSRoformer /
MDXCSeparator classes and against the CPU oracle.  Bar: 1e-4 relative RMS on stems."""
import os

import numpy as np
import pytest

from oracle import roformer_oracle as R

pytestmark = pytest.mark.gpu
TOL = 1e-4

CFG = R.RoformerConfig(dim=32, depth=2, heads=2, dim_head=64, freqs_per_bands=(2, 2, 4, 8, 17), stft_n_fft=64,
                       stft_hop_length=16, stft_win_length=64, dim_t=21, sample_rate=100, mlp_expansion_factor=2)
CFG2 = R.RoformerConfig(dim=32, depth=1, heads=2, dim_head=64, freqs_per_bands=(2, 2, 4, 8, 17), stft_n_fft=64,
                        stft_hop_length=16, stft_win_length=64, dim_t=21, sample_rate=100, num_stems=2,
                        time_transformer_depth=2, freq_transformer_depth=2, target_instrument=None)
CFG3 = R.RoformerConfig(dim=32, depth=1, heads=2, dim_head=64, freqs_per_bands=(2, 2, 4, 8, 17), stft_n_fft=64,
                        stft_hop_length=16, stft_win_length=48, dim_t=21, sample_rate=100, mlp_expansion_factor=2)


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


@pytest.fixture(scope="module")
def A():
    import audio_separator_amd as A
    return A


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "roformer_small.npz"))


def demixer(A, cfg, seed, overlap, max_batch=0):
    return A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0,
                          "secondary_stem_name": cfg.instruments[1]},
                         {"overlap": overlap}, state_dict=R.make_roformer_state(cfg, seed), max_batch=max_batch)


def test_forward_golden(A, g):
    w = (0.4 * np.random.default_rng(81).standard_normal((2, 2, 320))).astype(np.float32)
    y = demixer(A, CFG, 7, 8).engine.rof_forward(w)
    assert y.shape == (2, 1, 2, 320)
    assert rel_rms(y[:, 0], g["fwd1"]) < TOL, rel_rms(y[:, 0], g["fwd1"])
    y2 = demixer(A, CFG2, 8, 8).engine.rof_forward(w)
    assert rel_rms(y2, g["fwd2"]) < TOL, rel_rms(y2, g["fwd2"])


STFT_OPTS = {"norm": dict(stft_normalized=True), "hamming": dict(stft_window_fn="hamming_window"),
             "norm_blackman_win48": dict(stft_normalized=True, stft_window_fn="blackman_window", stft_win_length=48)}


@pytest.mark.parametrize("key", list(STFT_OPTS))
def test_stft_options_golden(A, golden_dir, key):
    """BSRoformer(stft_normalized=True / stft_window_fn=...) (bs_roformer.py:332-333, 384-386) at the ENGINE level, goldens written by
    the reference class constructed with those arguments (tests/golden/make_golden_roformer.py): `stft_normalized` scales the spectrum
    by n_fft^-1/2 on the way in and the synthesis window by n_fft^1/2 on the way out; the window function is evaluated with torch by
    the host (mdxc.stft_window_table) and handed over as a table (asx_set_stft_window).  (Through the reference's loader these options
    reach the class on the mel and the legacy paths only -- _create_bs_roformer forwards neither, roformer_loader.py:123-150 -- and the
    plugin mirrors that: test_host_files.)"""
    from audio_separator_amd.mdxc import stft_window_table
    go = np.load(os.path.join(golden_dir, "roformer_stft_options.npz"))
    opt = STFT_OPTS[key]
    cfg = R.RoformerConfig(dim=32, depth=1, heads=2, dim_head=64, freqs_per_bands=(2, 2, 4, 8, 17), stft_n_fft=64, stft_hop_length=16,
                           dim_t=21, sample_rate=100, mlp_expansion_factor=2, **{"stft_win_length": 64, **opt})
    wl = cfg.stft_win_length
    eng = A.Engine(A.MDXConfig(n_fft=64, hop_length=16, dim_f=33, segment_size=21, overlap=0.0, win_length=0 if wl == 64 else wl))
    eng.load_rof(A.RofConfig(dim=32, depth=1, heads=2, dim_head=64, num_stems=1, time_transformer_depth=1, freq_transformer_depth=1,
                             mlp_expansion_factor=2, mask_estimator_depth=2, freqs_per_bands=cfg.freqs_per_bands, n_out=2,
                             stft_normalized=bool(opt.get("stft_normalized", False))), R.make_roformer_state(cfg, 10))
    if "stft_window_fn" in opt:
        eng.set_stft_window(stft_window_table("torch." + opt["stft_window_fn"], wl, 64))
    w = (0.4 * np.random.default_rng(81).standard_normal((2, 2, 320))).astype(np.float32)
    y = eng.rof_forward(w)
    assert rel_rms(y[:, 0], go["fwd_" + key]) < TOL, rel_rms(y[:, 0], go["fwd_" + key])
    mix = (0.4 * np.random.default_rng(3090).standard_normal((2, 777))).astype(np.float32)
    chunk = 16 * 20
    out = eng.rof_demix(mix, min(int(2.5 * 100), chunk))
    assert rel_rms(out[0], go["demix_" + key]) < TOL, rel_rms(out[0], go["demix_" + key])
    if "stft_window_fn" in opt:                        # the option is not a no-op: the Hann default gives a different signal
        eng2 = A.Engine(A.MDXConfig(n_fft=64, hop_length=16, dim_f=33, segment_size=21, overlap=0.0, win_length=0 if wl == 64 else wl))
        eng2.load_rof(eng.rof_cfg, R.make_roformer_state(cfg, 10))
        assert rel_rms(eng2.rof_forward(w)[:, 0], go["fwd_" + key]) > 1e-2


@pytest.mark.parametrize("arith", ["f16x3", "bf16x6"])
def test_forward_golden_both_matrix_pipes(A, g, arith):
    """The same golden vector through the split-operand kernels (row GEMM tdf3_kernel + attention6_kernel; fp16 x 3 -- the default -- and
    bf16 x 6) and through the fp32-MFMA kernels (gemm_bf16x6 = 0), with proof of which attention kernel ran (library launch counters)."""
    w = (0.4 * np.random.default_rng(81).standard_normal((2, 2, 320))).astype(np.float32)
    eng = demixer(A, CFG, 7, 8).engine
    try:
        eng.set_option("gemm_bf16x6", 1)
        eng.set_option("gemm_f16x3", 1 if arith == "f16x3" else 0)
        n0, h0 = eng.counter("attn6_launches"), eng.counter("attn6h_launches")
        y6 = eng.rof_forward(w)
        assert eng.counter("attn6_launches") > n0, "attention6_kernel did not run"
        assert (eng.counter("attn6h_launches") > h0) == (arith == "f16x3"), "the other arithmetic of attention6_kernel ran"
        assert np.array_equal(y6, eng.rof_forward(w)), "two forwards differ"
        eng.set_option("gemm_bf16x6", 0)
        n1 = eng.counter("attn6_launches")
        y32 = eng.rof_forward(w)
        assert eng.counter("attn6_launches") == n1, "the fp32 run went through attention6_kernel"
    finally:
        eng.set_option("gemm_bf16x6", 1)
        eng.set_option("gemm_f16x3", 1)
    e6, e32 = rel_rms(y6[:, 0], g["fwd1"]), rel_rms(y32[:, 0], g["fwd1"])
    assert e6 < TOL and e32 < TOL, (e6, e32)
    assert e6 <= 2.0 * e32 + 1e-6, (e6, e32)             # fp32-grade, not a reduced-precision mode
    assert rel_rms(y6, y32) < 2e-5, rel_rms(y6, y32)


def test_forward_win_length_golden(A, g):
    """stft_win_length < stft_n_fft: the Hann window is zero padded to n_fft at both ends (torch.stft / istft), vector written by
    the reference BSRoformer (asx_mdx_config.win_length, ABI 3)."""
    w = (0.4 * np.random.default_rng(81).standard_normal((2, 2, 320))).astype(np.float32)
    y = demixer(A, CFG3, 9, 8).engine.rof_forward(w)
    assert rel_rms(y[:, 0], g["fwd3_win48"]) < TOL, rel_rms(y[:, 0], g["fwd3_win48"])


@pytest.mark.parametrize("name,n,ov", [("n1000_ov8", 1000, 8), ("n1000_ov2", 1000, 2), ("n320_ov1", 320, 1),
                                       ("n777_ov2", 777, 2.5)])
def test_demix_single_stem_golden(A, g, name, n, ov):
    mix = (0.4 * np.random.default_rng(90 + n).standard_normal((2, n))).astype(np.float32)
    out = demixer(A, CFG, 7, ov, max_batch=3).demix(mix)
    assert rel_rms(out["vocals"], g[f"demix1_{name}_primary"]) < TOL, rel_rms(out["vocals"], g[f"demix1_{name}_primary"])
    assert rel_rms(out["other"], g[f"demix1_{name}_secondary"]) < TOL


def test_demix_two_stems_golden(A, g):
    mix = (0.4 * np.random.default_rng(1090).standard_normal((2, 1000))).astype(np.float32)
    out = demixer(A, CFG2, 8, 2).demix(mix)
    got = np.stack([out[k] for k in CFG2.instruments])
    assert rel_rms(got, g["demix2"]) < TOL, rel_rms(got, g["demix2"])


def test_short_mix_is_rejected(A):
    with pytest.raises(A.AsxError):
        demixer(A, CFG, 7, 8).demix(np.zeros((2, 100), np.float32))


def test_ep317_shape_excerpt_vs_oracle(A):
    # the public ep_317 layout (dim 512-class heads of 64, 62 bands, n_fft 2048, hop 441) with reduced width / depth /
    # frames so that the CPU oracle finishes: sequence lengths straddle the 64-wide attention tiles (T = 161, 62 bands)
    cfg = R.RoformerConfig(dim=128, depth=2, heads=4, dim_head=64, freqs_per_bands=R.DEFAULT_FREQS_PER_BANDS,
                           dim_t=161, mlp_expansion_factor=2)
    sd = R.make_roformer_state(cfg, 2)
    n = 441 * 160 * 2 + 3000
    mix = (0.3 * np.random.default_rng(10).standard_normal((2, n))).astype(np.float32)
    dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0, "secondary_stem_name": "other"},
                       {"overlap": 1}, state_dict=sd)
    out = dm.demix(mix)
    ref = R.roformer_demix(mix, sd, cfg, overlap=1)
    e = rel_rms(out["vocals"], ref[0])
    print("ep317-shaped excerpt rel-RMS:", e)
    assert e < TOL, e
    assert dm.engine.rof_flops(1) > 0


def test_feed_forward_pair_image_ab(A):
    """The hidden activations of every feed-forward travel between its two linears as a pair image (option "gemm_pair_images", round 6): same
    excerpt with the option on and off -- the pair-image reader ran once per feed-forward, never with the option off, and the stems agree
    far inside the parity tolerance (the products are the same up to the exponent the parts carry)."""
    cfg = R.RoformerConfig(dim=128, depth=2, heads=4, dim_head=64, freqs_per_bands=R.DEFAULT_FREQS_PER_BANDS,
                           dim_t=161, mlp_expansion_factor=2)
    sd = R.make_roformer_state(cfg, 2)
    n = 441 * 160 + 1000
    mix = (0.3 * np.random.default_rng(11).standard_normal((2, n))).astype(np.float32)
    dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0, "secondary_stem_name": "other"},
                       {"overlap": 1}, state_dict=sd)
    eng = dm.engine
    try:
        eng.set_option("gemm_pair_images", 1)
    except A.AsxError as exc:
        pytest.skip(f"default library: {exc}")
    p0 = eng.counter("tdf3_pair_image_launches")
    on = dm.demix(mix)["vocals"]
    got = eng.counter("tdf3_pair_image_launches") - p0
    passes = -(-n // (441 * 160))                            # chunks of the excerpt, one pass each at this length (overlap 1)
    assert got > 0 and got % (2 * cfg.depth) == 0, (got, passes)   # one per feed-forward: depth x (time + frequency transformer) per pass
    eng.set_option("gemm_pair_images", 0)
    p0 = eng.counter("tdf3_pair_image_launches")
    off = dm.demix(mix)["vocals"]
    assert eng.counter("tdf3_pair_image_launches") == p0
    assert rel_rms(on, off) < 2e-6, rel_rms(on, off)


def test_batching_is_invisible(A):
    mix = (0.4 * np.random.default_rng(13).standard_normal((2, 1500))).astype(np.float32)
    outs = [demixer(A, CFG, 7, 2, max_batch=mb).demix(mix)["vocals"] for mb in (1, 2, 64)]
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


# ---- Mel-Band Roformer ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gm(golden_dir):
    return np.load(os.path.join(golden_dir, "melroformer_small.npz"))


def mel_cfgs():
    c1 = R.RoformerConfig.mel_config(dim=32, depth=2, heads=2, dim_head=64, num_bands=6, stft_n_fft=64, stft_hop_length=16,
                                     stft_win_length=64, dim_t=21, sample_rate=100, mlp_expansion_factor=2)
    c2 = R.RoformerConfig.mel_config(dim=32, depth=1, heads=2, dim_head=64, num_bands=8, stft_n_fft=64, stft_hop_length=16,
                                     stft_win_length=64, dim_t=21, sample_rate=100, num_stems=2, time_transformer_depth=2,
                                     freq_transformer_depth=1, target_instrument=None, mask_estimator_depth=2)
    return c1, c2


def test_mel_layout_matches_oracle(A):
    from audio_separator_amd.mdxc import mel_band_layout
    for args in ((44100, 2048, 60), (100, 64, 6), (44100, 2048, 64)):
        assert mel_band_layout(*args) == R.mel_band_layout(*args)


def test_mel_forward_golden(A, gm):
    c1, c2 = mel_cfgs()
    w = (0.4 * np.random.default_rng(181).standard_normal((2, 2, 320))).astype(np.float32)
    y = demixer(A, c1, 17, 2).engine.rof_forward(w)
    assert rel_rms(y[:, 0], gm["fwd1"]) < TOL, rel_rms(y[:, 0], gm["fwd1"])
    y2 = demixer(A, c2, 18, 8).engine.rof_forward(w)
    assert rel_rms(y2, gm["fwd2"]) < TOL, rel_rms(y2, gm["fwd2"])


def test_mel_demix_golden(A, gm):
    c1, c2 = mel_cfgs()
    mix = (0.4 * np.random.default_rng(2090).standard_normal((2, 1000))).astype(np.float32)
    out = demixer(A, c1, 17, 2, max_batch=3).demix(mix)
    assert rel_rms(out["vocals"], gm["demix1_primary"]) < TOL, rel_rms(out["vocals"], gm["demix1_primary"])
    assert rel_rms(out["other"], gm["demix1_secondary"]) < TOL
    out2 = demixer(A, c2, 18, 8).demix(mix)
    got = np.stack([out2[k] for k in c2.instruments])
    assert rel_rms(got, gm["demix2"]) < TOL, rel_rms(got, gm["demix2"])
