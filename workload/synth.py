"""Seeded synthetic benchmark workload of the MDX path: the song and ConvTDFNet weights with the reference
class's state_dict names and shapes (uvr_lib_v5/mdxnet.py:54-95).  Neutral ground between the product and the checker:
``bench.py`` builds its timed workload from here, ``oracle/mdx_oracle.py`` re-exports the same names for the tests.  Nothing
here computes anything the hot path computes.

All ``file:line`` citations are relative to ``/root/reference/audio_separator/separator/``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class NetDims:
    dim_c: int = 4
    dim_f: int = 3072
    dim_t: int = 256
    g: int = 48
    l: int = 3
    num_blocks: int = 11
    k: int = 3
    bn: int | None = 8        # 0: one Linear(f, f) in the TDF branch; None: no TDF branch (modules.py:52-70)
    bias: bool = False        # TDF Linear bias (kuielab configs train with bias=False)
    norm: str = "batch"       # "batch" = BatchNorm2d (optimizer 'rmsprop'), "group" = GroupNorm(2, c) ('adamw'), mdxnet.py:45-49

    @property
    def n(self) -> int:
        return self.num_blocks // 2


def _bn_init(gen: torch.Generator, c: int, prefix: str, sd: dict, group: bool = False):
    sd[prefix + ".weight"] = 0.8 + 0.4 * torch.rand(c, generator=gen)
    sd[prefix + ".bias"] = 0.1 * torch.randn(c, generator=gen)
    if not group:                                      # GroupNorm carries no running statistics
        sd[prefix + ".running_mean"] = 0.1 * torch.randn(c, generator=gen)
        sd[prefix + ".running_var"] = 0.5 + torch.rand(c, generator=gen)


def _tfc_tdf_init(gen, c, f, d: NetDims, prefix, sd):
    grp = d.norm == "group"
    for j in range(d.l):
        fan = c * d.k * d.k
        sd[f"{prefix}.tfc.H.{j}.0.weight"] = torch.randn(c, c, d.k, d.k, generator=gen) * math.sqrt(2.0 / fan)
        sd[f"{prefix}.tfc.H.{j}.0.bias"] = 0.05 * torch.randn(c, generator=gen)
        _bn_init(gen, c, f"{prefix}.tfc.H.{j}.1", sd, grp)
    if d.bn is None:
        return
    fb = f if d.bn == 0 else f // d.bn
    sd[f"{prefix}.tdf.0.weight"] = torch.randn(fb, f, generator=gen) * math.sqrt(1.0 / f)
    if d.bias:
        sd[f"{prefix}.tdf.0.bias"] = 0.05 * torch.randn(fb, generator=gen)
    _bn_init(gen, c, f"{prefix}.tdf.1", sd, grp)
    if d.bn == 0:
        return
    sd[f"{prefix}.tdf.3.weight"] = torch.randn(f, fb, generator=gen) * math.sqrt(1.0 / fb)
    if d.bias:
        sd[f"{prefix}.tdf.3.bias"] = 0.05 * torch.randn(f, generator=gen)
    _bn_init(gen, c, f"{prefix}.tdf.4", sd, grp)


def make_convtdf_state(d: NetDims, seed: int = 0) -> dict:
    """Seeded synthetic weights with the reference ConvTDFNet's state_dict
    names and shapes (mdxnet.py:54-95), BatchNorm running stats randomised so
    that folding is exercised.  Scales keep activations O(1) through the
    skip-multiplies of the decoder."""
    gen = torch.Generator().manual_seed(seed)
    sd: dict = {}
    g = d.g
    sd["first_conv.0.weight"] = torch.randn(g, d.dim_c, 1, 1, generator=gen) * math.sqrt(2.0 / d.dim_c)
    sd["first_conv.0.bias"] = 0.05 * torch.randn(g, generator=gen)
    _bn_init(gen, g, "first_conv.1", sd, d.norm == "group")
    f, c = d.dim_f, g
    for i in range(d.n):
        _tfc_tdf_init(gen, c, f, d, f"encoding_blocks.{i}", sd)
        sd[f"ds.{i}.0.weight"] = torch.randn(c + g, c, 2, 2, generator=gen) * math.sqrt(2.0 / (4 * c))
        sd[f"ds.{i}.0.bias"] = 0.05 * torch.randn(c + g, generator=gen)
        _bn_init(gen, c + g, f"ds.{i}.1", sd, d.norm == "group")
        f //= 2
        c += g
    _tfc_tdf_init(gen, c, f, d, "bottleneck_block", sd)
    for i in range(d.n):
        sd[f"us.{i}.0.weight"] = torch.randn(c, c - g, 2, 2, generator=gen) * math.sqrt(1.0 / c)
        sd[f"us.{i}.0.bias"] = 0.05 * torch.randn(c - g, generator=gen)
        _bn_init(gen, c - g, f"us.{i}.1", sd, d.norm == "group")
        f *= 2
        c -= g
        _tfc_tdf_init(gen, c, f, d, f"decoding_blocks.{i}", sd)
    sd["final_conv.0.weight"] = torch.randn(d.dim_c, c, 1, 1, generator=gen) * math.sqrt(1.0 / c)
    sd["final_conv.0.bias"] = 0.05 * torch.randn(d.dim_c, generator=gen)
    return {k: v.float().contiguous() for k, v in sd.items()}


def synth_mix(n_samples: int, seed: int = 0, sr: int = 44100) -> np.ndarray:
    """Seeded synthetic stereo input: 8 random sinusoids +
    0.1*N(0,1), peak 0.9.  float32 [2, n_samples]."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64) / sr
    out = np.zeros((2, n_samples), dtype=np.float64)
    for ch in range(2):
        for _ in range(8):
            f0 = rng.uniform(50.0, 8000.0)
            a = rng.uniform(0.1, 1.0)
            ph = rng.uniform(0, 2 * np.pi)
            out[ch] += a * np.sin(2 * np.pi * f0 * t + ph)
        out[ch] += 0.1 * rng.standard_normal(n_samples)
    out *= 0.9 / np.abs(out).max()
    return out.astype(np.float32)
