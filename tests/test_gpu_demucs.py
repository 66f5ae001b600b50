"""GPU parity of the Demucs v4 path against golden vectors written by the reference HTDemucs / apply_model classes
(tests/golden/make_golden_demucs.py) and against the CPU oracle.  Bar: 1e-4 relative RMS on the separated sources."""
import os
from fractions import Fraction

import numpy as np
import pytest

from oracle import demucs_oracle as D

pytestmark = pytest.mark.gpu
TOL = 1e-4


def ocfg_a():
    return D.HTConfig(channels=16, nfft=1024, depth=3, bottom_channels=128, t_layers=3, t_heads=2, samplerate=8000,
                      segment=Fraction(1, 1))


def ocfg_b():
    return D.HTConfig(channels=24, nfft=1024, depth=3, bottom_channels=0, t_layers=2, t_heads=2, samplerate=8000,
                      segment=Fraction(1, 1))


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
    return np.load(os.path.join(golden_dir, "demucs_small.npz"))


def hcfg(A, oc, max_batch=0):
    return A.HTConfig(sources=tuple(oc.sources), channels=oc.channels, growth=oc.growth, nfft=oc.nfft, depth=oc.depth,
                      bottom_channels=oc.bottom_channels, t_layers=oc.t_layers, t_heads=oc.t_heads,
                      t_hidden_scale=oc.t_hidden_scale, samplerate=oc.samplerate, segment=oc.segment,
                      freq_emb=oc.freq_emb, max_batch=max_batch)


def demixer(A, oc, seed, max_batch=0, **arch):
    return A.DemucsDemixer({"torch_device": 0}, arch, models=[(hcfg(A, oc, max_batch), D.make_ht_state(oc, seed))])


@pytest.mark.parametrize("tag,oc,seed", [("a", ocfg_a(), 11), ("b", ocfg_b(), 12)])
def test_forward_golden(A, g, tag, oc, seed):
    dm = demixer(A, oc, seed)
    dm._load(0)
    y = dm.engine.ht_forward(g[f"{tag}_fwd_in"])
    assert y.shape == g[f"{tag}_fwd_out"].shape
    assert rel_rms(y, g[f"{tag}_fwd_out"]) < TOL, rel_rms(y, g[f"{tag}_fwd_out"])
    tl = oc.training_length
    ys = dm.engine.ht_forward(g[f"{tag}_fwd_in"][:1, :, : tl - 777])
    assert rel_rms(ys, g[f"{tag}_short_out"]) < TOL, rel_rms(ys, g[f"{tag}_short_out"])


def test_forward_engine_tables(A, g):
    """sinusoidal tables built by the engine itself (no host torch tables): within 5e-4 of the reference"""
    oc = ocfg_a()
    import audio_separator_amd as A2
    e = A2.Engine(A2.MDXConfig(n_fft=1024, hop_length=256, dim_f=512, segment_size=8))
    e.load_ht(hcfg(A, oc), D.make_ht_state(oc, 11), pos_tables=False)
    y = e.ht_forward(g["a_fwd_in"])
    assert rel_rms(y, g["a_fwd_out"]) < 5e-4, rel_rms(y, g["a_fwd_out"])


@pytest.mark.parametrize("max_batch", [0, 3])
def test_apply_model_split_golden(A, g, max_batch):
    dm = demixer(A, ocfg_a(), 11, max_batch=max_batch)
    dm._load(0)
    out = dm.engine.ht_demix(g["a_mix"][0], shifts=0, overlap=0.25)
    assert rel_rms(out, g["a_split"][0]) < TOL, rel_rms(out, g["a_split"][0])


def test_apply_model_shifts_golden(A, g):
    dm = demixer(A, ocfg_a(), 11)
    dm._load(0)
    out = dm.engine.ht_demix(g["a_mix"][0], shifts=2, offsets=[int(o) for o in g["a_offsets"]], overlap=0.25)
    assert rel_rms(out, g["a_shift"][0]) < TOL, rel_rms(out, g["a_shift"][0])


@pytest.mark.parametrize("n,shifts,overlap", [(20923, 2, 0.25), (9000, 1, 0.5), (8000, 0, 0.1), (3001, 1, 0.25)])
def test_demix_demucs_oracle(A, n, shifts, overlap):
    """demix_demucs (standardise, shifts, split, swap) against the oracle on ragged lengths"""
    oc = ocfg_a()
    sd = D.make_ht_state(oc, 11)
    mix = (0.3 * np.random.default_rng(n).standard_normal((2, n)) + 0.05).astype(np.float32)
    offs = [int(o) for o in np.random.default_rng(n + 1).integers(0, oc.samplerate // 2 + 1, size=max(shifts, 1))][:shifts]
    want = D.demix_demucs(mix, sd, oc, shifts=shifts, overlap=overlap, offsets=offs)
    dm = demixer(A, oc, 11, shifts=shifts, overlap=overlap)
    got = dm.demix(mix, offsets=[offs])
    assert got.shape == want.shape
    assert rel_rms(got, want) < TOL, rel_rms(got, want)


def test_bag_of_models(A):
    """BagOfModels weighting (apply.py:169-196) on two members"""
    oc = ocfg_a()
    sds = [D.make_ht_state(oc, 11), D.make_ht_state(oc, 21)]
    w = [[1.0, 0.0, 2.0, 1.0], [0.0, 1.0, 1.0, 1.0]]
    mix = (0.3 * np.random.default_rng(5).standard_normal((2, 9000))).astype(np.float32)
    dm = A.DemucsDemixer({"torch_device": 0}, {"shifts": 0}, models=[(hcfg(A, oc), sd) for sd in sds], weights=w)
    got = dm.demix(mix)
    import torch
    t = torch.from_numpy(mix)
    ref = t.mean(0)
    sm = ((t - ref.mean()) / ref.std()).numpy()
    est = 0
    for sd, wi in zip(sds, w):
        fn = lambda x, sd=sd: D.ht_forward(x.numpy(), sd, oc)  # noqa: E731
        est = est + D.apply_model(fn, sm[None], oc, shifts=0).numpy()[0] * np.asarray(wi, np.float32)[:, None, None]
    est = est / np.sum(np.asarray(w, np.float32), axis=0)[:, None, None]
    est = est * float(ref.std()) + float(ref.mean())
    est[[0, 1]] = est[[1, 0]]
    assert rel_rms(got, est) < TOL, rel_rms(got, est)


def test_bag_device_combine_equals_host_combine(A, monkeypatch):
    """The bag path with everything in HBM (resident member engines, asx_ht_standardize_dev / asx_ht_bag_accumulate_dev /
    asx_ht_bag_finish_dev) against the same members combined in numpy on the host (round 2's path): float32 arithmetic in the
    reference's order on both sides -> identical arrays; and the member engines stay resident across calls."""
    import audio_separator_amd.demucs as DM
    oc = ocfg_a()
    sds = [D.make_ht_state(oc, 11), D.make_ht_state(oc, 21), D.make_ht_state(oc, 31)]
    w = [[1.0, 0.0, 2.0, 0.5], [0.0, 1.0, 1.0, 1.0], [0.25, 0.5, 0.0, 1.0]]
    offs = [[100, 3000], [2222, 17], [5, 3999]]
    mix = (0.3 * np.random.default_rng(9).standard_normal((2, 12345)) + 0.01).astype(np.float32)
    mk = lambda: A.DemucsDemixer({"torch_device": 0}, {"shifts": 2}, models=[(hcfg(A, oc), sd) for sd in sds], weights=w)  # noqa: E731
    dm = mk()
    got = dm.demix(mix, offsets=offs)
    engines = list(dm.engines)
    assert all(e is not None for e in engines)
    again = dm.demix(mix, offsets=offs)
    assert [id(e) for e in dm.engines] == [id(e) for e in engines] and np.array_equal(got, again)
    monkeypatch.setattr(DM, "_cuda_ready", lambda: False)
    host = mk().demix(mix, offsets=offs)
    assert got.shape == host.shape
    assert rel_rms(got, host) < 2e-7, rel_rms(got, host)


def test_error_paths(A):
    oc = ocfg_a()
    dm = demixer(A, oc, 11)
    with pytest.raises(ValueError):
        dm.demix(np.zeros((1, 100), np.float32))
    bad = D.HTConfig(channels=16, nfft=1024, depth=3, bottom_channels=96, t_layers=1, t_heads=8, samplerate=8000,
                     segment=Fraction(1, 1))   # head dim 12
    with pytest.raises(A.AsxError):
        demixer(A, bad, 1)._load(0)


def test_segments_disabled_matches_oracle(A):
    # apply_model(split=False): the shifted track is centred in a training-length window (valid_length) and trimmed; longer
    # inputs fail like the reference (HTDemucs.valid_length raises)
    oc = ocfg_a()
    rng = np.random.default_rng(9)
    mix = (rng.standard_normal((2, 3300)) * 0.2 + 0.01).astype(np.float32)
    offs = [[2100, 40]]
    dm = demixer(A, oc, 11, shifts=2, segments_enabled=False)
    out = dm.demix(mix, offsets=offs)
    ref = D.demix_demucs(mix, D.make_ht_state(oc, 11), oc, shifts=2, split=False, offsets=offs[0])
    assert rel_rms(out, ref) < TOL, rel_rms(out, ref)
    with pytest.raises(ValueError, match="training length"):
        dm.demix(np.zeros((2, 9000), np.float32) + mix[:, :1], offsets=offs)
