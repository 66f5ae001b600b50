"""Whole-song parity cases shared by tools/fullsong_oracle.py (CPU leg, runs anywhere) and tools/fullsong_parity.py (GPU leg).

Every case is fully determined by seeds: both legs rebuild the same synthetic weights and the same input from them, so the
only thing that travels between the legs is the oracle's output at a fixed set of sample WINDOWS (the full float stems of a
4-minute, 4-stem song are 340 MB; 16 windows of 32768 samples -- the first and the last anchored at the song's edges -- are
17 MB and see every chunk seam region of the song often enough to catch a fold error) plus whole-song statistics.

Realistic magnitudes: the synthetic ConvTDFNet has a gain of ~1e8; its final 1x1 conv is rescaled by a
calibration factor (one chunk through the oracle, target stem RMS 0.1) that is STORED with the oracle record and re-applied by
the GPU leg, so the 0.9 normalisation threshold, the 1e-6 silence threshold and the int16 quantisation operate in their real
range.  The Demucs / MDX23C / VR synthetic nets already produce O(0.1 .. 1) stems.
"""
from __future__ import annotations

import os
from fractions import Fraction

import numpy as np

SR = 44100
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.path.join(ROOT, "gpurun_cache", "fullsong")

VR_MP = {"bins": 768, "unstable_bins": 7, "reduction_bins": 668, "sr": 44100, "pre_filter_start": 740, "pre_filter_stop": 768,
         "band": {1: {"sr": 11025, "hl": 128, "n_fft": 1024, "crop_start": 0, "crop_stop": 186, "lpf_start": 37, "lpf_stop": 73, "res_type": "polyphase"},
                  2: {"sr": 11025, "hl": 128, "n_fft": 512, "crop_start": 4, "crop_stop": 185, "hpf_start": 36, "hpf_stop": 18, "lpf_start": 93, "lpf_stop": 185, "res_type": "polyphase"},
                  3: {"sr": 22050, "hl": 256, "n_fft": 512, "crop_start": 46, "crop_stop": 186, "hpf_start": 93, "hpf_stop": 46, "lpf_start": 164, "lpf_stop": 186, "res_type": "polyphase"},
                  4: {"sr": 44100, "hl": 512, "n_fft": 768, "crop_start": 121, "crop_stop": 382, "hpf_start": 138, "hpf_stop": 123, "res_type": "sinc_medium"}}}

CASES = {
    # name: (seconds, description)
    "mdx_hq3": (240.0, "BASELINE config 1: UVR-MDX-NET-Inst_HQ_3 geometry, whole MDXSeparator.separate array path (normalise, demix, "
                       "* peak, secondary = mix - compensate * primary) + the writer's int16 pass, final conv rescaled to stem RMS ~0.1"),
    "htdemucs": (240.0, "BASELINE config 2: htdemucs layout, 4 stems, shifts 2 (offsets 11025, 3000), overlap 0.25, segment 7.8 s"),
    "hdemucs_mmi": (240.0, "hdemucs_mmi layout (Demucs v3), 4 stems, shifts 2 (offsets 11025, 3000), overlap 0.25, 44-s chunks"),
    "vr_2hp": (10.0, "BASELINE config 0: 10 s, VR arch 123821 (2_HP-UVR shape) on the 4band_44100 parameters, window 512, both stems; "
                     "synthesis chain with the polyphase converter (the reference's ARM / MPS rule) on both sides"),
    "vr_2hp_sinc": (10.0, "the same clip with the sinc_fastest converter on both sides (the reference's Linux / x86 rule, the plugin default here; "
                          "restated libsamplerate algorithm: INTEGRATION.md \"VR resampler\")"),
    "mdx23c": (60.0, "MDX23C (TFC-TDF v3) default layout, 60 s, overlap 4"),
    "bs_roformer": (241.0, "BASELINE config 3: BS-Roformer ep_317 layout at its own size (dim 512, depth 12, 8 heads, 62 bands, n_fft 2048, hop 441, "
                           "8-s chunks), 4 min + 1 s so that the loop has 31 chunks with the last one re-anchored at N - chunk (Hamming-weighted "
                           "fold over an 87 % overlap with chunk 29: mdxc_separator.py:310-343), overlap 8 (step = chunk)"),
}
OFFSETS = [11025, 3000]


def windows(n: int, count: int = 16, width: int = 32768):
    """Start indices of the comparison windows: evenly spread, first at 0, last ending at n."""
    width = min(width, n)
    if n <= count * width:
        return np.array([0], np.int64), n
    starts = np.linspace(0, n - width, count).astype(np.int64)
    return starts, width


def take(x: np.ndarray, starts, width):
    """x [..., n] -> [..., len(starts), width]"""
    return np.stack([x[..., s:s + width] for s in starts], axis=-2)


def synth(n: int, seed: int):
    from oracle import mdx_oracle as O
    return O.synth_mix(n, seed=seed)


def mdx_state(scale: float | None = None):
    from oracle import mdx_oracle as O
    d = O.NetDims()
    sd = O.make_convtdf_state(d, seed=0)
    if scale is not None:
        for k in ("final_conv.0.weight", "final_conv.0.bias"):
            sd[k] = sd[k] * scale
    return d, sd


def ht_config():
    from oracle import demucs_oracle as D
    return D.HTConfig()


def segment_fraction():
    return Fraction(39, 5)


def roformer_config():
    from oracle import roformer_oracle as R
    return R.RoformerConfig(freqs_per_bands=R.DEFAULT_FREQS_PER_BANDS)   # dim 512, depth 12, 8 heads, T = 801, hop 441
