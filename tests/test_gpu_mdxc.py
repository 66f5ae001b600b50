"""GPU parity of the MDXC (TFC-TDF v3 / MDX23C) path against golden vectors written by the reference
classes and against the CPU oracle.  Bar: 1e-4 relative RMS on stems."""
import json
import os

import numpy as np
import pytest

from oracle import mdxc_oracle as M

pytestmark = pytest.mark.gpu
TOL = 1e-4

CFG2 = M.V3Config(n_fft=128, hop_length=16, dim_f=64, dim_t=16, num_subbands=2, num_scales=2, num_blocks_per_scale=2,
                  num_channels_model=8, growth=8, bottleneck_factor=4)
CFG1 = M.V3Config(n_fft=128, hop_length=16, dim_f=64, dim_t=16, num_subbands=2, num_scales=2, num_blocks_per_scale=1,
                  num_channels_model=8, growth=4, bottleneck_factor=2, target_instrument="Vocals", act="relu")


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
    return np.load(os.path.join(golden_dir, "mdxc_small.npz"))


def demixer(A, cfg, seed, overlap, seg=None, max_batch=0, pitch_shift=0):
    md = cfg.as_model_data()
    arch = {"overlap": overlap, "batch_size": 2, "pitch_shift": pitch_shift}
    if seg is not None:
        arch.update(segment_size=seg, override_model_segment_size=True)
    return A.MDXCDemixer({"model_data": md, "torch_device": 0, "secondary_stem_name": "Instrumental"}, arch,
                         state_dict=M.make_v3_state(cfg, seed), max_batch=max_batch)


def test_forward_golden(A, g):
    w = (0.4 * np.random.default_rng(61).standard_normal((2, 2, 240))).astype(np.float32)
    d2 = demixer(A, CFG2, 5, 4)
    y2 = d2.engine.v3_forward(w)
    assert y2.shape == (2, 2, 2, 240)
    assert rel_rms(y2, g["fwd2"]) < TOL, rel_rms(y2, g["fwd2"])
    d1 = demixer(A, CFG1, 6, 2)
    y1 = d1.engine.v3_forward(w)
    assert rel_rms(y1[:, 0], g["fwd1"]) < TOL, rel_rms(y1[:, 0], g["fwd1"])


@pytest.mark.parametrize("name,n", [("n3000", 3000), ("n100", 100), ("n241", 241)])
def test_demix_two_stem_golden(A, g, name, n):
    mix = (0.4 * np.random.default_rng(70 + n).standard_normal((2, n))).astype(np.float32)
    out = demixer(A, CFG2, 5, 4, max_batch=5).demix(mix)
    got = np.stack([out[k] for k in CFG2.instruments])
    assert rel_rms(got, g[f"demix2_{name}"]) < TOL, rel_rms(got, g[f"demix2_{name}"])


def test_demix_overlap8_segment_override_golden(A, g):
    mix = (0.4 * np.random.default_rng(3070).standard_normal((2, 3000))).astype(np.float32)
    out = demixer(A, CFG2, 5, 8, seg=12).demix(mix)
    got = np.stack([out[k] for k in CFG2.instruments])
    assert rel_rms(got, g["demix2_ov8_seg12"]) < TOL


def test_demix_single_target_residual_golden(A, g):
    mix = (0.4 * np.random.default_rng(3070).standard_normal((2, 3000))).astype(np.float32)
    out = demixer(A, CFG1, 6, 2).demix(mix)
    assert rel_rms(out["Vocals"], g["demix1_primary"]) < TOL
    assert rel_rms(out["Instrumental"], g["demix1_secondary"]) < TOL


def test_plan_matches_reference_arithmetic(A):
    d = demixer(A, CFG2, 5, 4)
    for n in (1, 100, 240, 241, 3000, 44100):
        for ov in (1, 2, 4, 8):
            ref = M.mdxc_plan(n, CFG2, ov)
            p = d.engine.mdxc_plan(n, ov)
            assert (p["chunk_size"], p["step"], p["pad"], p["trim"], p["padded_len"], p["n_chunks"]) == ref


def test_mdx23c_shape_excerpt_vs_oracle(A):
    # the public MDX23C layout (n_fft 8192, dim_f 4096, 4 subbands, InstanceNorm + GELU) with reduced width /
    # depth so the CPU oracle finishes: 2 chunks through STFT -> net -> iSTFT -> fold
    cfg = M.V3Config(n_fft=8192, hop_length=1024, dim_f=4096, dim_t=64, num_subbands=4, num_scales=3,
                     num_blocks_per_scale=2, num_channels_model=32, growth=32, bottleneck_factor=4)
    sd = M.make_v3_state(cfg, 1)
    n = 1024 * 63 + 5000
    mix = (0.3 * np.random.default_rng(9).standard_normal((2, n))).astype(np.float32)
    dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0}, {"overlap": 2}, state_dict=sd)
    out = dm.demix(mix)
    ref = M.mdxc_demix(mix, sd, cfg, overlap=2)
    got = np.stack([out[k] for k in cfg.instruments])
    e = rel_rms(got, ref)
    print("MDX23C-shaped excerpt rel-RMS:", e)
    assert e < TOL, e
    # the engine's closed-form FLOP counter (feeds the roofline figures) against FlopCounterMode on the reference class
    # (tests/golden/make_flops_fixture.py)
    with open(os.path.join(os.path.dirname(__file__), "golden", "flops.json")) as fh:
        want = json.load(fh)["mdx23c_excerpt"]
    assert abs(dm.engine.v3_flops(1) - want) <= 1e-9 * want, (dm.engine.v3_flops(1), want)


@pytest.mark.parametrize("semis", [2, -3])
def test_pitch_shift_round_trip_vs_oracle(A, semis):
    """mdxc_separator.py:230-243, 268-270, 417-419: the mix resampled to sr * 2^(-p / 12) (libsamplerate sinc_fastest, channel by
    channel), separated, every stem resampled back and padded / trimmed to the mix's length"""
    mix = (0.4 * np.random.default_rng(21).standard_normal((2, 5000))).astype(np.float32)
    d = demixer(A, CFG2, 5, 4, pitch_shift=semis)
    out = d.demix(mix)
    got = np.stack([out[k] for k in CFG2.instruments])
    ref = M.mdxc_demix_pitched(mix, M.make_v3_state(CFG2, 5), CFG2, overlap=4, pitch_shift=semis)
    assert got.shape == ref.shape == (len(CFG2.instruments), 2, 5000)
    assert rel_rms(got, ref) < TOL, rel_rms(got, ref)
    plain = np.stack(list(demixer(A, CFG2, 5, 4).demix(mix).values()))
    assert rel_rms(got, plain) > 1e-3          # and it really is a different computation


def test_batching_is_invisible(A):
    mix = (0.4 * np.random.default_rng(12).standard_normal((2, 4000))).astype(np.float32)
    outs = [np.stack(list(demixer(A, CFG2, 5, 4, max_batch=mb).demix(mix).values())) for mb in (1, 3, 64)]
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
