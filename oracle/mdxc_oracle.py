"""CPU oracle for the MDXC (MDX23C / TFC-TDF v3) demix path.

TEST INFRASTRUCTURE ONLY (see oracle/mdx_oracle.py for the rules).  Restates
``uvr_lib_v5/tfc_tdf_v3.py`` (TFC_TDF_net.forward :230-267 and its modules :84-148,
STFT :5-53) and the TFC branch of ``MDXCSeparator.demix``
(architectures/mdxc_separator.py:345-404) in numpy / torch-CPU fp32.

Parity status: PINNED against golden vectors written by the reference classes
themselves (tests/golden/make_golden_mdxc.py -> tests/golden/mdxc_*.npz).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

from .mdx_oracle import stft_forward, stft_inverse


@dataclass
class V3Config:
    """The fields of the model YAML that TFC_TDF_net reads (tfc_tdf_v3.py:151-214)."""
    n_fft: int = 8192
    hop_length: int = 1024
    dim_f: int = 4096
    dim_t: int = 256
    num_channels: int = 2
    num_subbands: int = 4
    num_scales: int = 5
    scale: tuple = (2, 2)
    num_blocks_per_scale: int = 2
    num_channels_model: int = 128      # config.model.num_channels
    growth: int = 128
    bottleneck_factor: int = 4
    norm: str = "InstanceNorm"
    act: str = "gelu"
    instruments: tuple = ("Vocals", "Instrumental")
    target_instrument: str | None = None

    @property
    def num_targets(self) -> int:
        return 1 if self.target_instrument else len(self.instruments)

    @property
    def dim_c(self) -> int:
        return self.num_subbands * self.num_channels * 2

    def as_model_data(self) -> dict:
        """The nested dict the reference wraps in ConfigDict (mdxc_separator.py:82)."""
        return {"audio": {"n_fft": self.n_fft, "hop_length": self.hop_length, "dim_f": self.dim_f,
                          "num_channels": self.num_channels, "chunk_size": self.hop_length * (self.dim_t - 1)},
                "model": {"act": self.act, "norm": self.norm, "bottleneck_factor": self.bottleneck_factor,
                          "growth": self.growth, "num_blocks_per_scale": self.num_blocks_per_scale,
                          "num_channels": self.num_channels_model, "num_scales": self.num_scales,
                          "num_subbands": self.num_subbands, "scale": list(self.scale)},
                "training": {"instruments": list(self.instruments), "target_instrument": self.target_instrument},
                "inference": {"dim_t": self.dim_t, "batch_size": 1, "num_overlap": 4}}


def make_v3_state(cfg: V3Config, seed: int = 0) -> dict:
    """Seeded synthetic weights with the reference module's state_dict names/shapes."""
    gen = torch.Generator().manual_seed(seed)
    sd: dict = {}

    def norm(prefix, c):
        if cfg.norm:
            sd[prefix + ".weight"] = 0.8 + 0.4 * torch.rand(c, generator=gen)
            sd[prefix + ".bias"] = 0.1 * torch.randn(c, generator=gen)

    def tfc_tdf(prefix, in_c, c, f):
        for j in range(cfg.num_blocks_per_scale):
            p = f"{prefix}.blocks.{j}"
            norm(p + ".tfc1.0", in_c)
            sd[p + ".tfc1.2.weight"] = torch.randn(c, in_c, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * in_c))
            norm(p + ".tdf.0", c)
            sd[p + ".tdf.2.weight"] = torch.randn(f // cfg.bottleneck_factor, f, generator=gen) * math.sqrt(2.0 / f)
            norm(p + ".tdf.3", c)
            sd[p + ".tdf.5.weight"] = torch.randn(f, f // cfg.bottleneck_factor, generator=gen) * math.sqrt(
                2.0 / (f // cfg.bottleneck_factor))
            norm(p + ".tfc2.0", c)
            sd[p + ".tfc2.2.weight"] = torch.randn(c, c, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * c))
            sd[p + ".shortcut.weight"] = torch.randn(c, in_c, 1, 1, generator=gen) * math.sqrt(1.0 / in_c)
            in_c = c

    c, g, f = cfg.num_channels_model, cfg.growth, cfg.dim_f // cfg.num_subbands
    sd["first_conv.weight"] = torch.randn(c, cfg.dim_c, 1, 1, generator=gen) * math.sqrt(1.0 / cfg.dim_c)
    for i in range(cfg.num_scales):
        tfc_tdf(f"encoder_blocks.{i}.tfc_tdf", c, c, f)
        norm(f"encoder_blocks.{i}.downscale.conv.0", c)
        sd[f"encoder_blocks.{i}.downscale.conv.2.weight"] = torch.randn(c + g, c, *cfg.scale, generator=gen) * math.sqrt(
            2.0 / (c * cfg.scale[0] * cfg.scale[1]))
        f //= cfg.scale[1]
        c += g
    tfc_tdf("bottleneck_block", c, c, f)
    for i in range(cfg.num_scales):
        norm(f"decoder_blocks.{i}.upscale.conv.0", c)
        sd[f"decoder_blocks.{i}.upscale.conv.2.weight"] = torch.randn(c, c - g, *cfg.scale, generator=gen) * math.sqrt(1.0 / c)
        f *= cfg.scale[1]
        c -= g
        tfc_tdf(f"decoder_blocks.{i}.tfc_tdf", 2 * c, c, f)
    sd["final_conv.0.weight"] = torch.randn(c, c + cfg.dim_c, 1, 1, generator=gen) * math.sqrt(1.0 / (c + cfg.dim_c))
    sd["final_conv.2.weight"] = torch.randn(cfg.num_targets * cfg.dim_c, c, 1, 1, generator=gen) * math.sqrt(1.0 / c)
    return {k: v.float().contiguous() for k, v in sd.items()}


def _norm_act(x, sd, prefix, cfg: V3Config):
    if cfg.norm == "InstanceNorm":
        x = F.instance_norm(x, weight=sd[prefix + ".weight"], bias=sd[prefix + ".bias"], eps=1e-5)
    elif cfg.norm == "BatchNorm":
        raise NotImplementedError("BatchNorm variant needs running stats")
    elif cfg.norm and "GroupNorm" in cfg.norm:
        x = F.group_norm(x, int(cfg.norm.replace("GroupNorm", "")), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)
    if cfg.act == "gelu":
        return F.gelu(x)
    if cfg.act == "relu":
        return F.relu(x)
    raise NotImplementedError(cfg.act)


def _act(x, cfg):
    return F.gelu(x) if cfg.act == "gelu" else F.relu(x)


def _tfc_tdf(x, sd, prefix, cfg: V3Config):
    """TFC_TDF.forward (tfc_tdf_v3.py:141-148)."""
    for j in range(cfg.num_blocks_per_scale):
        p = f"{prefix}.blocks.{j}"
        s = F.conv2d(x, sd[p + ".shortcut.weight"])
        x = F.conv2d(_norm_act(x, sd, p + ".tfc1.0", cfg), sd[p + ".tfc1.2.weight"], padding=1)
        t = F.linear(_norm_act(x, sd, p + ".tdf.0", cfg), sd[p + ".tdf.2.weight"])
        t = F.linear(_norm_act(t, sd, p + ".tdf.3", cfg), sd[p + ".tdf.5.weight"])
        x = x + t
        x = F.conv2d(_norm_act(x, sd, p + ".tfc2.0", cfg), sd[p + ".tfc2.2.weight"], padding=1)
        x = x + s
    return x


@torch.no_grad()
def v3_core(spec, sd: dict, cfg: V3Config):
    """TFC_TDF_net.forward between the STFT and the iSTFT (tfc_tdf_v3.py:234-261).
    spec [B, 2*num_channels, dim_f, T] -> [B, (num_targets,) 2*num_channels, dim_f, T] (numpy in/out)."""
    x = torch.as_tensor(np.ascontiguousarray(spec), dtype=torch.float32)
    k = cfg.num_subbands
    b, c, f, t = x.shape
    x = x.reshape(b, c, k, f // k, t).reshape(b, c * k, f // k, t)            # cac2cws :216-221
    mix = x
    first = x = F.conv2d(x, sd["first_conv.weight"])
    x = x.transpose(-1, -2)
    enc = []
    for i in range(cfg.num_scales):
        x = _tfc_tdf(x, sd, f"encoder_blocks.{i}.tfc_tdf", cfg)
        enc.append(x)
        x = F.conv2d(_norm_act(x, sd, f"encoder_blocks.{i}.downscale.conv.0", cfg),
                     sd[f"encoder_blocks.{i}.downscale.conv.2.weight"], stride=tuple(cfg.scale))
    x = _tfc_tdf(x, sd, "bottleneck_block", cfg)
    for i in range(cfg.num_scales):
        x = F.conv_transpose2d(_norm_act(x, sd, f"decoder_blocks.{i}.upscale.conv.0", cfg),
                               sd[f"decoder_blocks.{i}.upscale.conv.2.weight"], stride=tuple(cfg.scale))
        x = torch.cat([x, enc.pop()], 1)
        x = _tfc_tdf(x, sd, f"decoder_blocks.{i}.tfc_tdf", cfg)
    x = x.transpose(-1, -2)
    x = x * first
    x = F.conv2d(torch.cat([mix, x], 1), sd["final_conv.0.weight"])
    x = F.conv2d(_act(x, cfg), sd["final_conv.2.weight"])
    b, c, f, t = x.shape
    x = x.reshape(b, c // k, k, f, t).reshape(b, c // k, f * k, t)            # cws2cac :223-228
    if cfg.num_targets > 1:
        x = x.reshape(b, cfg.num_targets, -1, f * k, t)
    return x.numpy()


def v3_forward(wave: np.ndarray, sd: dict, cfg: V3Config) -> np.ndarray:
    """TFC_TDF_net.forward (tfc_tdf_v3.py:230-267): wave [B,2,chunk] -> [B,(S,)2,chunk]."""
    spec = stft_forward(np.asarray(wave, np.float32), cfg.n_fft, cfg.hop_length, cfg.dim_f)
    y = v3_core(spec, sd, cfg)
    if cfg.num_targets > 1:
        b, s, c, f, t = y.shape
        out = stft_inverse(y.reshape(b * s, c, f, t), cfg.n_fft, cfg.hop_length)
        return out.reshape(b, s, 2, -1)
    return stft_inverse(y, cfg.n_fft, cfg.hop_length)


def mdxc_plan(n: int, cfg: V3Config, overlap: int, segment_size: int | None = None):
    """Index arithmetic of the TFC branch (mdxc_separator.py:354-372)."""
    seg = segment_size if segment_size is not None else cfg.dim_t
    chunk_size = cfg.hop_length * (seg - 1)
    hop_size = chunk_size // overlap
    pad_size = hop_size - (n - chunk_size) % hop_size          # Python floor-mod
    front = chunk_size - hop_size
    total = front + n + pad_size + chunk_size - hop_size
    n_chunks = (total - chunk_size) // hop_size + 1            # torch.Tensor.unfold
    return chunk_size, hop_size, pad_size, front, total, n_chunks


def mdxc_demix(mix: np.ndarray, sd: dict, cfg: V3Config, overlap: int = 8, segment_size: int | None = None):
    """MDXCSeparator.demix, TFC branch (mdxc_separator.py:345-404): [2,N] -> [S,2,N] (S>1) or [2,N]."""
    mix = np.asarray(mix, np.float32)
    n = mix.shape[1]
    chunk_size, hop_size, pad_size, front, total, n_chunks = mdxc_plan(n, cfg, overlap, segment_size)
    padded = np.concatenate([np.zeros((2, front), np.float32), mix,
                             np.zeros((2, pad_size + chunk_size - hop_size), np.float32)], 1)
    S = cfg.num_targets
    acc = np.zeros((S, 2, total), np.float32) if S > 1 else np.zeros((2, total), np.float32)
    for k in range(n_chunks):
        chunk = padded[:, k * hop_size: k * hop_size + chunk_size]
        out = v3_forward(chunk[None], sd, cfg)[0]
        acc[..., k * hop_size: k * hop_size + chunk_size] += out
    return acc[..., front: -(pad_size + chunk_size - hop_size)] / overlap


# --------------------------------------------------------------------------
# pitch shift around the demix call (mdxc_separator.py:230-243, 268-270, 417-419; spec_utils.change_pitch_semitones :783-790,
# match_array_shapes :752-769).  On Linux / x86 `wav_resolution_float_resampling` is "sinc_fastest" (spec_utils.py:33-38):
# librosa.resample -> libsamplerate, channel by channel -- restated (parity unpinned) in oracle/vr_oracle.py.
# --------------------------------------------------------------------------
def change_pitch_semitones(y: np.ndarray, sr, semitone_shift):
    from oracle import vr_oracle as V
    factor = 2 ** (semitone_shift / 12)
    out = np.array([V.lr_resample(np.asarray(ch), orig_sr=sr, target_sr=sr * factor, res_type="sinc_fastest") for ch in y])
    return out, sr * factor


def match_array_shapes(a1: np.ndarray, a2: np.ndarray):
    if a1.shape[1] > a2.shape[1]:
        return a1[:, : a2.shape[1]]
    if a1.shape[1] < a2.shape[1]:
        return np.pad(a1, ((0, 0), (0, a2.shape[1] - a1.shape[1])), "constant", constant_values=0)
    return a1


def mdxc_demix_pitched(mix: np.ndarray, sd: dict, cfg: V3Config, overlap: int, pitch_shift: int, sample_rate=44100):
    """MDXCSeparator.demix with pitch_shift != 0, multi-stem TFC branch: resample the mix to sr * 2^(-p/12), demix, resample every
    stem back by 2^(p/12) and pad / trim to the original length."""
    mix = np.asarray(mix, np.float32)
    mixp, srp = change_pitch_semitones(mix, sample_rate, -pitch_shift)
    stems = mdxc_demix(mixp.astype(np.float32), sd, cfg, overlap=overlap)
    if stems.ndim == 2:
        stems = stems[None]
    return np.stack([match_array_shapes(change_pitch_semitones(s_, srp, pitch_shift)[0], mix) for s_ in stems])
