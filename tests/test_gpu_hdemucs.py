"""GPU parity of the Demucs v3 (HDemucs) path against golden vectors written by the reference HDemucs / apply_model classes
(tests/golden/make_golden_hdemucs.py) and against the CPU oracle.  Bar: 1e-4 relative RMS on the separated sources."""
import os

import numpy as np
import pytest

from oracle import hdemucs_oracle as H

pytestmark = pytest.mark.gpu
TOL = 1e-4


def ocfg():
    return H.HDConfig(channels=8, nfft=1024, depth=5, norm_starts=3, dconv_attn=3, dconv_lstm=3, samplerate=8000, segment=2)


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
    return np.load(os.path.join(golden_dir, "hdemucs_small.npz"))


def hcfg(A, oc, max_batch=0):
    return A.HDConfig(sources=tuple(oc.sources), channels=oc.channels, growth=oc.growth, nfft=oc.nfft, depth=oc.depth,
                      norm_starts=oc.norm_starts, dconv_attn=oc.dconv_attn, dconv_lstm=oc.dconv_lstm, samplerate=oc.samplerate,
                      segment=oc.segment, freq_emb=oc.freq_emb, max_batch=max_batch)


def demixer(A, oc=None, seed=21, max_batch=0, **arch):
    oc = oc or ocfg()
    return A.DemucsDemixer({"torch_device": 0}, arch, models=[(hcfg(A, oc, max_batch), H.make_hd_state(oc, seed))])


@pytest.fixture(scope="module")
def dm(A):
    d = demixer(A)
    d._load(0)
    return d


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f"])
def test_forward_golden(dm, g, tag):
    # a: 47 frames (plain BLSTM); b: 219 frames (overlapped BLSTM frames, demucs.py:41-64); c: odd length, batch 2;
    # d, e, f: 877 / 100 / 3 samples -- shorter than nfft and than the reflect pad (pad1d's zero extension), 4 / 1 / 1 frames
    y = dm.engine.hd_forward(g[f"x_{tag}"])
    assert y.shape == g[f"y_{tag}"].shape
    assert rel_rms(y, g[f"y_{tag}"]) < TOL, rel_rms(y, g[f"y_{tag}"])


def test_forward_wide_hidden_golden(A, g):
    # DConv hidden 64 / 128 -> LocalState head dims 16 / 32: the MFMA flash-attention path with the decay slope, 79 frames
    oc = H.HDConfig(channels=8, nfft=1024, depth=5, norm_starts=3, dconv_attn=3, dconv_lstm=3, dconv_comp=1, samplerate=8000, segment=2)
    hc = A.HDConfig(sources=tuple(oc.sources), channels=8, nfft=1024, depth=5, norm_starts=3, dconv_attn=3, dconv_lstm=3, dconv_comp=1,
                    samplerate=8000, segment=2)
    d = A.DemucsDemixer({"torch_device": 0}, {}, models=[(hc, H.make_hd_state(oc, 22))])
    d._load(0)
    y = d.engine.hd_forward(g["x_w"])
    assert rel_rms(y, g["y_w"]) < TOL, rel_rms(y, g["y_w"])


def test_forward_lengths_share_one_engine(dm, g):
    # the workspace is re-planned per length; going back to an earlier length gives the same numbers (float64 atomics in
    # the statistics make the last bits order-dependent)
    y1 = dm.engine.hd_forward(g["x_c"])
    dm.engine.hd_forward(g["x_a"])
    y2 = dm.engine.hd_forward(g["x_c"])
    assert rel_rms(y1, y2) < 1e-6


@pytest.mark.parametrize("max_batch", [0, 1])
def test_apply_model_split_golden(A, g, max_batch):
    d = demixer(A, max_batch=max_batch)
    d._load(0)
    out = d.engine.hd_demix(g["mix"][0], shifts=0, overlap=0.25)
    assert rel_rms(out, g["split"][0]) < TOL, rel_rms(out, g["split"][0])


def test_apply_model_shifts_golden(dm, g):
    out = dm.engine.hd_demix(g["mix"][0], shifts=2, offsets=[int(o) for o in g["offsets"]], overlap=0.25)
    assert rel_rms(out, g["shift"][0]) < TOL, rel_rms(out, g["shift"][0])


def test_demix_demucs_matches_oracle(A):
    # demix_demucs framing: standardise by the mono reference, apply_model, de-standardise, swap stems 0 / 1
    oc = ocfg()
    rng = np.random.default_rng(5)
    mix = (rng.standard_normal((2, 30011)) * 0.2 + 0.01).astype(np.float32)
    offs = [[1234, 77]]
    d = demixer(A, shifts=2, overlap=0.25)
    out = d.demix(mix, offsets=offs)
    ref = H.demix_hdemucs(mix, H.make_hd_state(oc, 21), oc, shifts=2, overlap=0.25, offsets=offs[0])
    assert rel_rms(out, ref) < TOL, rel_rms(out, ref)


def test_sharded_halves_match_single_call(dm, g):
    import torch
    mix = torch.from_numpy(g["mix"][0]).cuda()
    n = mix.shape[1]
    offs = [int(o) for o in g["offsets"]]
    e = dm.engine
    plan = e.hd_plan(n, shifts=2, offsets=offs)
    S, k, c = 4, plan["n_chunks"], plan["chunk_size"]
    chunks = torch.zeros(k, S, 2, c, device="cuda")
    mid = k // 2
    e.hd_segments_dev(mix.data_ptr(), n, 0, mid, chunks.data_ptr(), shifts=2, offsets=offs)
    e.hd_segments_dev(mix.data_ptr(), n, mid, k, chunks[mid:].data_ptr(), shifts=2, offsets=offs)
    out = torch.empty(S, 2, n, device="cuda")
    e.hd_fold_dev(mix.data_ptr(), n, chunks.data_ptr(), out.data_ptr(), shifts=2, offsets=offs)
    torch.cuda.synchronize()
    assert rel_rms(out.cpu().numpy(), g["shift"][0]) < TOL


def test_unsupported_structure_is_rejected(A):
    oc = ocfg()
    with pytest.raises(NotImplementedError):
        A.hdconfig_from_kwargs(dict(sources=list(oc.sources), depth=6, dconv_lstm=5))
    with pytest.raises(NotImplementedError):
        A.hdconfig_from_kwargs(dict(sources=list(oc.sources), hybrid_old=True))


def test_segments_disabled_matches_oracle(A):
    # segments_enabled=False -> apply_model(split=False): one forward per shift over the whole shifted track (apply.py:251-260)
    oc = ocfg()
    rng = np.random.default_rng(8)
    mix = (rng.standard_normal((2, 23011)) * 0.2 - 0.02).astype(np.float32)
    offs = [[3999, 5]]
    d = demixer(A, shifts=2, overlap=0.25, segments_enabled=False)
    out = d.demix(mix, offsets=offs)
    ref = H.demix_hdemucs(mix, H.make_hd_state(oc, 21), oc, shifts=2, split=False, offsets=offs[0])
    assert rel_rms(out, ref) < TOL, rel_rms(out, ref)


def test_v3_then_v4_on_one_engine(A, g):
    # loading a v4 net replaces the strided levels the v3 net was using: the v3 entry points must refuse, not run on them
    from fractions import Fraction
    from oracle import demucs_oracle as D
    d = demixer(A)
    d._load(0)
    e = d.engine
    y = e.hd_forward(g["x_c"])
    assert rel_rms(y, g["y_c"]) < TOL
    oc = D.HTConfig(channels=16, nfft=1024, depth=3, bottom_channels=128, t_layers=1, t_heads=2, samplerate=8000, segment=Fraction(1, 1))
    e.load_ht(A.HTConfig(sources=tuple(oc.sources), channels=16, nfft=1024, depth=3, bottom_channels=128, t_layers=1, t_heads=2,
                         samplerate=8000, segment=Fraction(1, 1)), D.make_ht_state(oc, 11))
    with pytest.raises(RuntimeError, match="not committed"):
        e.hd_forward(g["x_c"])
    e.load_hd(hcfg(A, ocfg()), H.make_hd_state(ocfg(), 21))
    with pytest.raises(RuntimeError, match="not committed"):
        e.ht_forward(np.zeros((1, 2, 8000), np.float32))
    assert rel_rms(e.hd_forward(g["x_c"]), g["y_c"]) < TOL


def test_joint_recurrences_across_lengths(A):
    # 8-s chunks at 8 kHz are 250 frames: the two full chunks and the second chunk of each shift (235 / 231 frames) are framed BLSTM
    # sequences of 200 steps in three groups of different length that share the recurrence launches at level A
    # (hd_forward_groups); the short tails (47 / 43 frames) and everything at level Z (125 / 118 / 116 ... frames) run their own
    # unframed recurrences -- five groups advance together
    oc = H.HDConfig(channels=8, nfft=1024, depth=5, norm_starts=3, dconv_attn=3, dconv_lstm=3, samplerate=8000, segment=8)
    hc = A.HDConfig(sources=tuple(oc.sources), channels=8, nfft=1024, depth=5, norm_starts=3, dconv_attn=3, dconv_lstm=3, samplerate=8000,
                    segment=8)
    sd = H.make_hd_state(oc, 21)
    d = A.DemucsDemixer({"torch_device": 0}, {"shifts": 2, "overlap": 0.25}, models=[(hc, sd)])
    mix = (0.25 * np.random.default_rng(12).standard_normal((2, 105000)) + 0.01).astype(np.float32)
    offs = [[1000, 2000]]
    out = d.demix(mix, offsets=offs)
    ref = H.demix_hdemucs(mix, sd, oc, shifts=2, overlap=0.25, offsets=offs[0])
    assert rel_rms(out, ref) < TOL, rel_rms(out, ref)
