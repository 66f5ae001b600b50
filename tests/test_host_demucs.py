"""Host logic of the Demucs mirror (python-audio-separator_amd/demucs.py) without a GPU: the engine is replaced by a stand-in that
answers the C-ABI calls with the CPU oracle, so what is tested is the Python side -- bag weighting (apply.py:169-196),
segments_enabled=False windowing (apply.py:198-214, 251-260), standardisation / stem swap, config mapping and error behaviour."""
import os
import sys
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import audio_separator_amd as A  # noqa: E402
from oracle import demucs_oracle as D  # noqa: E402
from oracle import hdemucs_oracle as H  # noqa: E402

HOC = H.HDConfig(channels=8, nfft=1024, depth=5, norm_starts=3, dconv_attn=3, dconv_lstm=3, samplerate=8000, segment=2)
TOC = D.HTConfig(channels=16, nfft=1024, depth=3, bottom_channels=128, t_layers=1, t_heads=2, samplerate=8000, segment=Fraction(1, 1))


class OracleEngine:
    """the Engine methods DemucsDemixer uses, answered by the oracle"""

    def load_hd(self, hc, sd):
        self.kind, self.sd = "hd", sd

    def load_ht(self, hc, sd):
        self.kind, self.sd = "ht", sd

    def hd_forward(self, x):
        return H.hd_forward(x, self.sd, HOC)

    def ht_forward(self, x):
        return D.ht_forward(x, self.sd, TOC)

    def _demix(self, mix, shifts=0, offsets=None, overlap=0.25, standardize=False, swap01=False):
        import torch
        cfg, fwd = (HOC, self.hd_forward) if self.kind == "hd" else (TOC, self.ht_forward)
        m = torch.tensor(mix)
        ref = m.mean(0)
        if standardize:
            m = (m - ref.mean()) / ref.std()
        fn = lambda x: fwd(x.numpy() if hasattr(x, "numpy") else x)  # noqa: E731
        out = D.apply_model(fn, m[None], cfg, shifts=shifts, split=True, overlap=overlap, offsets=offsets)[0]
        if standardize:
            out = out * ref.std() + ref.mean()
        out = out.numpy().astype(np.float32) if hasattr(out, "numpy") else np.asarray(out, np.float32)
        if swap01:
            out[[0, 1]] = out[[1, 0]]
        return out

    hd_demix = ht_demix = _demix


def _demixer(models, weights=None, **arch):
    d = A.DemucsDemixer({"torch_device": 0}, arch, models=models, weights=weights)
    d.engine = OracleEngine()
    return d


def _hd(seed):
    return (A.HDConfig(sources=tuple(HOC.sources), channels=8, nfft=1024, depth=5, norm_starts=3, dconv_attn=3, dconv_lstm=3, samplerate=8000,
                       segment=2), H.make_hd_state(HOC, seed))


def _ht(seed):
    return (A.HTConfig(sources=tuple(TOC.sources), channels=16, nfft=1024, depth=3, bottom_channels=128, t_layers=1, t_heads=2, samplerate=8000,
                       segment=Fraction(1, 1)), D.make_ht_state(TOC, seed))


def _close(a, b, tol=1e-5):
    return np.abs(np.asarray(a, np.float64) - b).max() <= tol * np.abs(b).max()


def test_single_model_goes_through_one_call():
    mix = (0.2 * np.random.default_rng(1).standard_normal((2, 21000)) + 0.01).astype(np.float32)
    d = _demixer([_hd(21)], shifts=1)
    out = d.demix(mix, offsets=[[1234]])
    want = H.demix_hdemucs(mix, H.make_hd_state(HOC, 21), HOC, shifts=1, offsets=[1234])
    assert _close(out, want)


def test_bag_of_models_weighting():
    # BagOfModels (apply.py:169-196): every model on the standardised mix with its own shift draws, per-source weights, sum / totals,
    # then demix_demucs' de-standardisation and stem swap (demucs_separator.py:183-187)
    import torch
    mix = (0.2 * np.random.default_rng(2).standard_normal((2, 9000)) - 0.02).astype(np.float32)
    w = [[1.0, 0.0, 0.5, 2.0], [0.0, 1.0, 0.5, 1.0]]
    offs = [[100, 3000], [2222, 17]]
    d = _demixer([_ht(11), _ht(12)], weights=w, shifts=2)
    out = d.demix(mix, offsets=offs)
    m = torch.tensor(mix)
    ref = m.mean(0)
    ms = ((m - ref.mean()) / ref.std())[None]
    est, tot = 0, np.zeros(4)
    for i, seed in enumerate((11, 12)):
        sd = D.make_ht_state(TOC, seed)
        o = D.apply_model(lambda x: D.ht_forward(x.numpy() if hasattr(x, "numpy") else x, sd, TOC), ms, TOC, shifts=2, offsets=offs[i])[0].numpy()
        est = est + o * np.asarray(w[i], np.float32)[:, None, None]
        tot += w[i]
    est = est / tot[:, None, None] * float(ref.std()) + float(ref.mean())
    est[[0, 1]] = est[[1, 0]]
    assert _close(out, est)


@pytest.mark.parametrize("kind", ["hd", "ht"])
def test_segments_disabled_windowing(kind):
    if kind == "hd":
        mix = (0.2 * np.random.default_rng(3).standard_normal((2, 15011)) + 0.01).astype(np.float32)
        d = _demixer([_hd(21)], shifts=2, segments_enabled=False)
        want = H.demix_hdemucs(mix, H.make_hd_state(HOC, 21), HOC, shifts=2, split=False, offsets=[3999, 5])
        out = d.demix(mix, offsets=[[3999, 5]])
    else:
        mix = (0.2 * np.random.default_rng(4).standard_normal((2, 3300)) + 0.01).astype(np.float32)
        d = _demixer([_ht(11)], shifts=2, segments_enabled=False)
        want = D.demix_demucs(mix, D.make_ht_state(TOC, 11), TOC, shifts=2, split=False, offsets=[2100, 40])
        out = d.demix(mix, offsets=[[2100, 40]])
    assert _close(out, want)


def test_segments_disabled_refuses_long_input_for_v4():
    d = _demixer([_ht(11)], shifts=0, segments_enabled=False)
    with pytest.raises(ValueError, match="training length"):
        d.demix(np.ones((2, 9000), np.float32) * 0.1 + np.arange(9000, dtype=np.float32)[None] * 1e-5)


def test_argument_errors_match_the_reference_style():
    d = _demixer([_hd(21)])
    with pytest.raises(ValueError, match="2-channel"):
        d.demix(np.zeros((1, 1000), np.float32))
    with pytest.raises(ValueError):
        A.DemucsDemixer({"torch_device": 0}, {}, models=[])
    with pytest.raises(ValueError, match="weights"):
        A.DemucsDemixer({"torch_device": 0}, {}, models=[_hd(21)], weights=[[1.0, 1.0]])


def test_config_from_checkpoint_kwargs():
    k = dict(sources=["drums", "bass", "other", "vocals"], channels=48, depth=6, nfft=4096, segment=44, samplerate=44100)
    hc = A.hdconfig_from_kwargs(k, max_batch=3)
    assert (hc.depth, hc.norm_starts, hc.dconv_lstm, hc.dconv_attn, hc.dconv_comp, hc.segment_samples, hc.max_batch) == (6, 4, 4, 4, 4, 44 * 44100, 3)
    for bad in (dict(dconv_lstm=5), dict(hybrid_old=True), dict(cac=False), dict(multi_freqs=[0.5]), dict(norm_starts=3)):
        with pytest.raises(NotImplementedError):
            A.hdconfig_from_kwargs(dict(k, **bad))
    tk = dict(sources=["drums", "bass", "other", "vocals"], segment=Fraction(39, 5))
    tc = A.htconfig_from_kwargs(tk)
    assert tc.segment_samples == 343980 and tc.t_layers == 5
    with pytest.raises(NotImplementedError):
        A.htconfig_from_kwargs(dict(tk, t_sparse_self_attn=True))
