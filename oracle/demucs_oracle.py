"""CPU oracle for the Demucs v4 (HTDemucs) demix path.

TEST INFRASTRUCTURE ONLY (see oracle/mdx_oracle.py for the rules).  Restates, in torch-CPU fp32:
``uvr_lib_v5/demucs/htdemucs.py`` (HTDemucs.forward :483-620, _spec :383, _ispec :405, _magnitude
:415, _mask :426), ``hdemucs.py`` (HEncLayer :67-170, HDecLayer :252-330, ScaledEmbedding :37,
pad1d :21), ``demucs.py`` (DConv :99, LayerScale :85), ``transformer.py`` (CrossTransformerEncoder
:415-560, MyTransformerEncoderLayer :189, CrossTransformerEncoderLayer :268, create_sin_embedding
:18, create_2d_sin_embedding :27), ``spec.py`` (spectro / ispectro), ``apply.py`` (apply_model
:124-260, TensorChunk :71, center_trim utils.py:53) and ``DemucsSeparator.demix_demucs``
(architectures/demucs_separator.py:162-194).

Supported configuration = the default HTDemucs structure: all `depth` layers are frequency layers
with kernel 8 / stride 4 (nfft/2 / 4^depth > 1), DConv in the encoder only (dconv_mode=1, depth 2),
no GroupNorm inside the encoder/decoder layers (norm_starts >= depth), rewrite convs, context 1 /
context_enc 0, CaC, sinusoidal embeddings, norm_first transformer with LayerScale and norm_out.

Parity status: PINNED on golden vectors written by the reference HTDemucs / apply_model classes
(tests/golden/make_golden_demucs.py -> demucs_small.npz).
"""
from __future__ import annotations

import math
import random
from dataclasses import dataclass
from fractions import Fraction

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class HTConfig:
    sources: tuple = ("drums", "bass", "other", "vocals")
    audio_channels: int = 2
    channels: int = 48
    growth: int = 2
    nfft: int = 4096
    depth: int = 4
    kernel_size: int = 8
    stride: int = 4
    dconv_depth: int = 2
    dconv_comp: int = 8
    freq_emb: float = 0.2
    bottom_channels: int = 0
    t_layers: int = 5
    t_heads: int = 8
    t_hidden_scale: float = 4.0
    samplerate: int = 44100
    segment: Fraction = Fraction(39, 5)

    @property
    def hop(self):
        return self.nfft // 4

    @property
    def training_length(self):
        return int(self.segment * self.samplerate)

    def ctor_kwargs(self) -> dict:
        return dict(sources=list(self.sources), audio_channels=self.audio_channels, channels=self.channels,
                    growth=self.growth, nfft=self.nfft, depth=self.depth, kernel_size=self.kernel_size,
                    stride=self.stride, dconv_depth=self.dconv_depth, dconv_comp=self.dconv_comp,
                    freq_emb=self.freq_emb, bottom_channels=self.bottom_channels, t_layers=self.t_layers,
                    t_heads=self.t_heads, t_hidden_scale=self.t_hidden_scale, samplerate=self.samplerate,
                    segment=self.segment)

    def chans(self):
        """(chin, chout) per layer for the freq (z) and time branches (htdemucs.py:250-330)."""
        S = len(self.sources)
        z, t = [], []
        chin, chin_z = self.audio_channels, self.audio_channels * 2
        chout = chout_z = self.channels
        for i in range(self.depth):
            z.append((chin_z, chout_z))
            t.append((chin, chout))
            if i == 0:
                chin = self.audio_channels * S
                chin_z = chin * 2
            dec_z = (chout_z, chin_z)
            dec_t = (chout, chin)
            z[-1] = z[-1] + dec_z
            t[-1] = t[-1] + dec_t
            chin, chin_z = chout, chout_z
            chout, chout_z = int(self.growth * chout), int(self.growth * chout_z)
        return z, t


def make_ht_state(cfg: HTConfig, seed: int = 0) -> dict:
    """Seeded synthetic weights with HTDemucs' state_dict names and shapes."""
    gen = torch.Generator().manual_seed(seed)
    sd: dict = {}

    def rn(*shape, scale=1.0):
        return torch.randn(*shape, generator=gen) * scale

    def dconv(prefix, ch):
        hidden = int(ch / cfg.dconv_comp)
        for d in range(cfg.dconv_depth):
            p = f"{prefix}.layers.{d}"
            sd[f"{p}.0.weight"] = rn(hidden, ch, 3, scale=math.sqrt(1.0 / (3 * ch)))
            sd[f"{p}.0.bias"] = rn(hidden, scale=0.05)
            sd[f"{p}.1.weight"] = 0.8 + 0.4 * torch.rand(hidden, generator=gen)
            sd[f"{p}.1.bias"] = rn(hidden, scale=0.1)
            sd[f"{p}.3.weight"] = rn(2 * ch, hidden, 1, scale=math.sqrt(1.0 / hidden))
            sd[f"{p}.3.bias"] = rn(2 * ch, scale=0.05)
            sd[f"{p}.4.weight"] = 0.8 + 0.4 * torch.rand(2 * ch, generator=gen)
            sd[f"{p}.4.bias"] = rn(2 * ch, scale=0.1)
            sd[f"{p}.6.scale"] = 0.3 + 0.4 * torch.rand(ch, generator=gen)

    zc, tc = cfg.chans()
    K = cfg.kernel_size
    for i in range(cfg.depth):
        ci, co, dci, dco = zc[i]
        sd[f"encoder.{i}.conv.weight"] = rn(co, ci, K, 1, scale=math.sqrt(2.0 / (K * ci)))
        sd[f"encoder.{i}.conv.bias"] = rn(co, scale=0.05)
        sd[f"encoder.{i}.rewrite.weight"] = rn(2 * co, co, 1, 1, scale=math.sqrt(2.0 / co))
        sd[f"encoder.{i}.rewrite.bias"] = rn(2 * co, scale=0.05)
        dconv(f"encoder.{i}.dconv", co)
        ti, to, dti, dto = tc[i]
        sd[f"tencoder.{i}.conv.weight"] = rn(to, ti, K, scale=math.sqrt(2.0 / (K * ti)))
        sd[f"tencoder.{i}.conv.bias"] = rn(to, scale=0.05)
        sd[f"tencoder.{i}.rewrite.weight"] = rn(2 * to, to, 1, scale=math.sqrt(2.0 / to))
        sd[f"tencoder.{i}.rewrite.bias"] = rn(2 * to, scale=0.05)
        dconv(f"tencoder.{i}.dconv", to)
        j = cfg.depth - 1 - i          # decoders are inserted at the front
        sd[f"decoder.{j}.conv_tr.weight"] = rn(dci, dco, K, 1, scale=math.sqrt(1.0 / dci))
        sd[f"decoder.{j}.conv_tr.bias"] = rn(dco, scale=0.05)
        sd[f"decoder.{j}.rewrite.weight"] = rn(2 * dci, dci, 3, 3, scale=math.sqrt(2.0 / (9 * dci)))
        sd[f"decoder.{j}.rewrite.bias"] = rn(2 * dci, scale=0.05)
        sd[f"tdecoder.{j}.conv_tr.weight"] = rn(dti, dto, K, scale=math.sqrt(1.0 / dti))
        sd[f"tdecoder.{j}.conv_tr.bias"] = rn(dto, scale=0.05)
        sd[f"tdecoder.{j}.rewrite.weight"] = rn(2 * dti, dti, 3, scale=math.sqrt(2.0 / (3 * dti)))
        sd[f"tdecoder.{j}.rewrite.bias"] = rn(2 * dti, scale=0.05)
    freqs0 = cfg.nfft // 2 // cfg.stride
    sd["freq_emb.embedding.weight"] = rn(freqs0, cfg.channels, scale=0.05)
    C = cfg.channels * cfg.growth ** (cfg.depth - 1)
    if cfg.bottom_channels:
        for n in ("channel_upsampler", "channel_upsampler_t"):
            sd[f"{n}.weight"] = rn(cfg.bottom_channels, C, 1, scale=math.sqrt(1.0 / C))
            sd[f"{n}.bias"] = rn(cfg.bottom_channels, scale=0.05)
        for n in ("channel_downsampler", "channel_downsampler_t"):
            sd[f"{n}.weight"] = rn(C, cfg.bottom_channels, 1, scale=math.sqrt(1.0 / cfg.bottom_channels))
            sd[f"{n}.bias"] = rn(C, scale=0.05)
        C = cfg.bottom_channels
    if cfg.t_layers:
        hid = int(C * cfg.t_hidden_scale)

        def ln(name):
            sd[name + ".weight"] = 0.8 + 0.4 * torch.rand(C, generator=gen)
            sd[name + ".bias"] = rn(C, scale=0.1)
        for br in ("norm_in", "norm_in_t"):
            ln(f"crosstransformer.{br}")
        for br in ("layers", "layers_t"):
            for i in range(cfg.t_layers):
                p = f"crosstransformer.{br}.{i}"
                attn = "self_attn" if i % 2 == 0 else "cross_attn"
                sd[f"{p}.{attn}.in_proj_weight"] = rn(3 * C, C, scale=math.sqrt(1.0 / C))
                sd[f"{p}.{attn}.in_proj_bias"] = rn(3 * C, scale=0.05)
                sd[f"{p}.{attn}.out_proj.weight"] = rn(C, C, scale=math.sqrt(1.0 / C))
                sd[f"{p}.{attn}.out_proj.bias"] = rn(C, scale=0.05)
                sd[f"{p}.linear1.weight"] = rn(hid, C, scale=math.sqrt(1.0 / C))
                sd[f"{p}.linear1.bias"] = rn(hid, scale=0.05)
                sd[f"{p}.linear2.weight"] = rn(C, hid, scale=math.sqrt(1.0 / hid))
                sd[f"{p}.linear2.bias"] = rn(C, scale=0.05)
                for nm in (("norm1", "norm2") if i % 2 == 0 else ("norm1", "norm2", "norm3")):
                    ln(f"{p}.{nm}")
                ln(f"{p}.norm_out")
                sd[f"{p}.gamma_1.scale"] = 0.3 + 0.4 * torch.rand(C, generator=gen)
                sd[f"{p}.gamma_2.scale"] = 0.3 + 0.4 * torch.rand(C, generator=gen)
    return {k: v.float().contiguous() for k, v in sd.items()}


# --------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------
def _dconv(x, sd, prefix, cfg):
    """DConv.forward (demucs.py:176-179): x [N, C, T]."""
    for d in range(cfg.dconv_depth):
        p = f"{prefix}.layers.{d}"
        dil = 2 ** d
        y = F.conv1d(x, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], dilation=dil, padding=dil)
        y = F.gelu(F.group_norm(y, 1, sd[f"{p}.1.weight"], sd[f"{p}.1.bias"]))
        y = F.conv1d(y, sd[f"{p}.3.weight"], sd[f"{p}.3.bias"])
        y = F.glu(F.group_norm(y, 1, sd[f"{p}.4.weight"], sd[f"{p}.4.bias"]), dim=1)
        x = x + sd[f"{p}.6.scale"][:, None] * y
    return x


def _enc_freq(x, sd, p, cfg, inject=None):
    """HEncLayer.forward, freq=True (hdemucs.py:139-170)."""
    y = F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=(cfg.stride, 1), padding=(cfg.kernel_size // 4, 0))
    if inject is not None:
        y = y + inject
    y = F.gelu(y)
    B, C, Fr, T = y.shape
    y = _dconv(y.permute(0, 2, 1, 3).reshape(-1, C, T), sd, p + ".dconv", cfg)
    y = y.view(B, Fr, C, T).permute(0, 2, 1, 3)
    return F.glu(F.conv2d(y, sd[p + ".rewrite.weight"], sd[p + ".rewrite.bias"]), dim=1)


def _enc_time(x, sd, p, cfg):
    """HEncLayer.forward, freq=False."""
    le = x.shape[-1]
    if le % cfg.stride:
        x = F.pad(x, (0, cfg.stride - (le % cfg.stride)))
    y = F.gelu(F.conv1d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=cfg.stride, padding=cfg.kernel_size // 4))
    y = _dconv(y, sd, p + ".dconv", cfg)
    return F.glu(F.conv1d(y, sd[p + ".rewrite.weight"], sd[p + ".rewrite.bias"]), dim=1)


def _dec_freq(x, skip, sd, p, cfg, last):
    """HDecLayer.forward, freq=True (hdemucs.py:303-330); returns (z, pre)."""
    x = x + skip
    y = F.glu(F.conv2d(x, sd[p + ".rewrite.weight"], sd[p + ".rewrite.bias"], padding=1), dim=1)
    z = F.conv_transpose2d(y, sd[p + ".conv_tr.weight"], sd[p + ".conv_tr.bias"], stride=(cfg.stride, 1))
    pad = cfg.kernel_size // 4
    z = z[..., pad:-pad, :]
    return (z if last else F.gelu(z)), y


def _dec_time(x, skip, sd, p, cfg, last, length):
    x = x + skip
    y = F.glu(F.conv1d(x, sd[p + ".rewrite.weight"], sd[p + ".rewrite.bias"], padding=1), dim=1)
    z = F.conv_transpose1d(y, sd[p + ".conv_tr.weight"], sd[p + ".conv_tr.bias"], stride=cfg.stride)
    pad = cfg.kernel_size // 4
    z = z[..., pad:pad + length]
    return z if last else F.gelu(z)


def sin_embedding(length, dim, max_period=10000.0):
    """create_sin_embedding (transformer.py:18-25), shift 0 -> [length, dim]."""
    pos = torch.arange(length).view(-1, 1)
    half = dim // 2
    adim = torch.arange(half).view(1, -1)
    phase = pos / (max_period ** (adim / (half - 1)))
    return torch.cat([torch.cos(phase), torch.sin(phase)], dim=-1)


def sin_embedding_2d(d_model, height, width, max_period=10000.0):
    """create_2d_sin_embedding (transformer.py:27-46) -> [d_model, height, width]."""
    pe = torch.zeros(d_model, height, width)
    dm = d_model // 2
    div = torch.exp(torch.arange(0.0, dm, 2) * -(math.log(max_period) / dm))
    pw = torch.arange(0.0, width).unsqueeze(1)
    ph = torch.arange(0.0, height).unsqueeze(1)
    pe[0:dm:2] = torch.sin(pw * div).transpose(0, 1).unsqueeze(1).repeat(1, height, 1)
    pe[1:dm:2] = torch.cos(pw * div).transpose(0, 1).unsqueeze(1).repeat(1, height, 1)
    pe[dm::2] = torch.sin(ph * div).transpose(0, 1).unsqueeze(2).repeat(1, 1, width)
    pe[dm + 1::2] = torch.cos(ph * div).transpose(0, 1).unsqueeze(2).repeat(1, 1, width)
    return pe


def _mha(q_in, kv_in, sd, p, heads):
    """nn.MultiheadAttention(batch_first=True), need_weights=False: [B, N, C]."""
    C = q_in.shape[-1]
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(q_in, w[:C], b[:C])
    k = F.linear(kv_in, w[C:2 * C], b[C:2 * C])
    v = F.linear(kv_in, w[2 * C:], b[2 * C:])
    B, Nq, _ = q.shape
    Nk = k.shape[1]
    dh = C // heads
    q = q.view(B, Nq, heads, dh).transpose(1, 2)
    k = k.view(B, Nk, heads, dh).transpose(1, 2)
    v = v.view(B, Nk, heads, dh).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-2, -1)) / math.sqrt(dh), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, Nq, C)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _gn_tc(x, sd, p):
    """MyGroupNorm(1, C) on [B, T, C] (transformer.py:181-187)."""
    return F.group_norm(x.transpose(1, 2), 1, sd[p + ".weight"], sd[p + ".bias"], 1e-5).transpose(1, 2)


def _ff(x, sd, p):
    return F.linear(F.gelu(F.linear(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                    sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])


def _self_layer(x, sd, p, heads):
    """MyTransformerEncoderLayer.forward, norm_first, layer_scale, norm_out (transformer.py:255-266)."""
    xn = _ln(x, sd, p + ".norm1")
    x = x + sd[p + ".gamma_1.scale"] * _mha(xn, xn, sd, p + ".self_attn", heads)
    x = x + sd[p + ".gamma_2.scale"] * _ff(_ln(x, sd, p + ".norm2"), sd, p)
    return _gn_tc(x, sd, p + ".norm_out")


def _cross_layer(q, k, sd, p, heads):
    """CrossTransformerEncoderLayer.forward (transformer.py:372-396)."""
    x = q + sd[p + ".gamma_1.scale"] * _mha(_ln(q, sd, p + ".norm1"), _ln(k, sd, p + ".norm2"), sd, p + ".cross_attn", heads)
    x = x + sd[p + ".gamma_2.scale"] * _ff(_ln(x, sd, p + ".norm3"), sd, p)
    return _gn_tc(x, sd, p + ".norm_out")


def _cross_transformer(x, xt, sd, cfg):
    """CrossTransformerEncoder.forward (transformer.py:520-548)."""
    B, C, Fr, T1 = x.shape
    pe2 = sin_embedding_2d(C, Fr, T1).permute(2, 1, 0).reshape(1, T1 * Fr, C)        # b (t1 fr) c
    x = x.permute(0, 3, 2, 1).reshape(B, T1 * Fr, C)
    x = _ln(x, sd, "crosstransformer.norm_in") + pe2
    T2 = xt.shape[-1]
    xt = xt.permute(0, 2, 1)
    xt = _ln(xt, sd, "crosstransformer.norm_in_t") + sin_embedding(T2, C)[None]
    for i in range(cfg.t_layers):
        if i % 2 == 0:
            x = _self_layer(x, sd, f"crosstransformer.layers.{i}", cfg.t_heads)
            xt = _self_layer(xt, sd, f"crosstransformer.layers_t.{i}", cfg.t_heads)
        else:
            old = x
            x = _cross_layer(x, xt, sd, f"crosstransformer.layers.{i}", cfg.t_heads)
            xt = _cross_layer(xt, old, sd, f"crosstransformer.layers_t.{i}", cfg.t_heads)
    x = x.reshape(B, T1, Fr, C).permute(0, 3, 2, 1)
    return x, xt.permute(0, 2, 1)


def _pad1d_reflect(x, left, right):
    """hdemucs.pad1d(mode='reflect') (hdemucs.py:21-34)."""
    length = x.shape[-1]
    mx = max(left, right)
    if length <= mx:
        extra = mx - length + 1
        er = min(right, extra)
        el = extra - er
        x = F.pad(x, (el, er))
        left, right = left - el, right - er
    return F.pad(x, (left, right), mode="reflect")


def _spec(x, cfg):
    hl = cfg.hop
    le = int(math.ceil(x.shape[-1] / hl))
    pad = hl // 2 * 3
    x = _pad1d_reflect(x, pad, pad + le * hl - x.shape[-1])
    B, C, L = x.shape
    z = torch.stft(x.reshape(-1, L), cfg.nfft, hl, window=torch.hann_window(cfg.nfft), win_length=cfg.nfft,
                   normalized=True, center=True, return_complex=True, pad_mode="reflect")
    z = z.view(B, C, z.shape[-2], z.shape[-1])[..., :-1, :]
    return z[..., 2:2 + le]


def _ispec(z, length, cfg):
    hl = cfg.hop
    z = F.pad(F.pad(z, (0, 0, 0, 1)), (2, 2))
    pad = hl // 2 * 3
    le = hl * int(math.ceil(length / hl)) + 2 * pad
    shp = z.shape
    x = torch.istft(z.reshape(-1, shp[-2], shp[-1]), cfg.nfft, hl, window=torch.hann_window(cfg.nfft),
                    win_length=cfg.nfft, normalized=True, length=le, center=True)
    x = x.view(*shp[:-2], le)
    return x[..., pad:pad + length]


@torch.no_grad()
def ht_forward(mix, sd: dict, cfg: HTConfig):
    """HTDemucs.forward (htdemucs.py:483-620), eval, use_train_segment: [B, 2, L] -> [B, S, 2, L]."""
    mix = torch.as_tensor(np.ascontiguousarray(mix), dtype=torch.float32)
    length_pre_pad = None
    tl = cfg.training_length
    if mix.shape[-1] < tl:
        length_pre_pad = mix.shape[-1]
        mix = F.pad(mix, (0, tl - length_pre_pad))
    z = _spec(mix, cfg)
    B, C, Fq, T = z.shape
    x = torch.view_as_real(z).permute(0, 1, 4, 2, 3).reshape(B, C * 2, Fq, T)
    mean = x.mean(dim=(1, 2, 3), keepdim=True)
    std = x.std(dim=(1, 2, 3), keepdim=True)
    x = (x - mean) / (1e-5 + std)
    xt = mix
    meant = xt.mean(dim=(1, 2), keepdim=True)
    stdt = xt.std(dim=(1, 2), keepdim=True)
    xt = (xt - meant) / (1e-5 + stdt)
    saved, saved_t, lengths_t = [], [], []
    for i in range(cfg.depth):
        lengths_t.append(xt.shape[-1])
        xt = _enc_time(xt, sd, f"tencoder.{i}", cfg)
        saved_t.append(xt)
        x = _enc_freq(x, sd, f"encoder.{i}", cfg)
        if i == 0 and cfg.freq_emb:
            frs = torch.arange(x.shape[-2])
            emb = (F.embedding(frs, sd["freq_emb.embedding.weight"]) * 10.0).t()[None, :, :, None].expand_as(x)
            x = x + cfg.freq_emb * emb
        saved.append(x)
    if cfg.t_layers:
        if cfg.bottom_channels:
            b, c, f, t = x.shape
            x = F.conv1d(x.reshape(b, c, f * t), sd["channel_upsampler.weight"], sd["channel_upsampler.bias"]).view(b, -1, f, t)
            xt = F.conv1d(xt, sd["channel_upsampler_t.weight"], sd["channel_upsampler_t.bias"])
        x, xt = _cross_transformer(x, xt, sd, cfg)
        if cfg.bottom_channels:
            b, c, f, t = x.shape
            x = F.conv1d(x.reshape(b, c, f * t), sd["channel_downsampler.weight"], sd["channel_downsampler.bias"]).view(b, -1, f, t)
            xt = F.conv1d(xt, sd["channel_downsampler_t.weight"], sd["channel_downsampler_t.bias"])
    for j in range(cfg.depth):
        last = j == cfg.depth - 1
        x, _ = _dec_freq(x, saved.pop(-1), sd, f"decoder.{j}", cfg, last)
        xt = _dec_time(xt, saved_t.pop(-1), sd, f"tdecoder.{j}", cfg, last, lengths_t.pop(-1))
    S = len(cfg.sources)
    x = x.view(B, S, -1, Fq, T) * std[:, None] + mean[:, None]
    zout = torch.view_as_complex(x.view(B, S, -1, 2, Fq, T).permute(0, 1, 2, 4, 5, 3).contiguous())
    x = _ispec(zout, tl, cfg)
    xt = xt.view(B, S, -1, tl) * stdt[:, None] + meant[:, None]
    x = xt + x
    if length_pre_pad:
        x = x[..., :length_pre_pad]
    return x.numpy()


# --------------------------------------------------------------------------
# apply_model (apply.py:124-260) and demix_demucs (demucs_separator.py:162-194)
# --------------------------------------------------------------------------
def _padded(tensor, offset, length, target):
    """TensorChunk(tensor, offset, length).padded(target) (apply.py:71-107): real context from `tensor`
    around the chunk, zeros only beyond the tensor's ends."""
    total = tensor.shape[-1]
    length = min(total - offset, length)
    delta = target - length
    start = offset - delta // 2
    end = start + target
    cs, ce = max(0, start), min(total, end)
    return F.pad(tensor[..., cs:ce], (cs - start, end - ce)), length


def _center_trim(t, ref):
    delta = t.shape[-1] - ref
    return t[..., delta // 2: t.shape[-1] - (delta - delta // 2)] if delta else t


def _valid_length(cfg, length):
    """apply.py:251-255: HTDemucs pads every leaf call to its training length; HDemucs (no valid_length) runs at `length`."""
    tl = getattr(cfg, "training_length", None)
    return tl if tl else length


def apply_split(model_fn, tensor, base, length, cfg: HTConfig, overlap=0.25):
    """The `split` branch of apply_model (apply.py:215-250) + the leaf call (:251-260) on the view
    tensor[..., base:base+length] (a TensorChunk of a TensorChunk keeps the parent tensor, so the
    per-chunk padding sees real audio outside the view)."""
    batch, channels = tensor.shape[:2]
    S = len(cfg.sources)
    out = torch.zeros(batch, S, channels, length)
    sum_weight = torch.zeros(length)
    segment = int(cfg.samplerate * cfg.segment)
    stride = int((1 - overlap) * segment)
    weight = torch.cat([torch.arange(1, segment // 2 + 1), torch.arange(segment - segment // 2, 0, -1)])
    weight = (weight / weight.max()) ** 1.0
    for offset in range(0, length, stride):
        clen = min(length - offset, segment)
        padded, _ = _padded(tensor, base + offset, clen, _valid_length(cfg, clen))
        chunk_out = _center_trim(torch.as_tensor(model_fn(padded)), clen)
        out[..., offset:offset + segment] += weight[:clen] * chunk_out
        sum_weight[offset:offset + segment] += weight[:clen]
    return out / sum_weight


def _apply_view(model_fn, tensor, base, length, cfg, split, overlap):
    if split:
        return apply_split(model_fn, tensor, base, length, cfg, overlap)
    padded, _ = _padded(tensor, base, length, _valid_length(cfg, length))
    return _center_trim(torch.as_tensor(model_fn(padded)), length)


def apply_model(model_fn, mix, cfg: HTConfig, shifts=1, split=True, overlap=0.25, offsets=None):
    """apply_model for a single model; `offsets` replaces the random.randint draws (apply.py:202-214)."""
    mix = torch.as_tensor(np.ascontiguousarray(mix), dtype=torch.float32)
    batch, channels, length = mix.shape
    if shifts:
        max_shift = int(0.5 * cfg.samplerate)
        padded, _ = _padded(mix, 0, length, length + 2 * max_shift)
        out = 0
        for i in range(shifts):
            offset = offsets[i] if offsets is not None else random.randint(0, max_shift)
            so = _apply_view(model_fn, padded, offset, length + max_shift - offset, cfg, split, overlap)
            out = out + so[..., max_shift - offset:]
        return out / shifts
    return _apply_view(model_fn, mix, 0, length, cfg, split, overlap)


def demix_demucs(mix: np.ndarray, sd: dict, cfg: HTConfig, shifts=2, overlap=0.25, split=True, offsets=None) -> np.ndarray:
    """DemucsSeparator.demix_demucs (demucs_separator.py:162-194): [2, N] -> [S, 2, N] (stems 0/1 swapped)."""
    m = torch.tensor(np.asarray(mix, np.float32))
    ref = m.mean(0)
    m = (m - ref.mean()) / ref.std()
    fn = lambda x: ht_forward(x.numpy() if hasattr(x, "numpy") else x, sd, cfg)  # noqa: E731
    src = apply_model(fn, m[None], cfg, shifts=shifts, split=split, overlap=overlap, offsets=offsets)[0]
    src = (src * ref.std() + ref.mean()).numpy()
    src[[0, 1]] = src[[1, 0]]
    return src


# --------------------------------------------------------------------------
# segment-list form of demix_demucs (test double of asx_ht_plan / asx_ht_segments_dev / asx_ht_fold_dev)
# --------------------------------------------------------------------------
def segment_plan(n: int, cfg: HTConfig, shifts, offsets, overlap=0.25):
    """[(shift index, view base, view length, chunk offset, chunk length)] in the reference's order"""
    seg = cfg.training_length
    stride = int((1 - overlap) * seg)
    max_shift = int(0.5 * cfg.samplerate) if shifts else 0
    plan = []
    for si in range(max(shifts, 1)):
        off = offsets[si] if shifts else 0
        vl = n + max_shift - off
        for o in range(0, vl, stride):
            plan.append((si, off, vl, o, min(vl - o, seg)))
    return plan, stride, max_shift


def _standardized(mix):
    m = torch.tensor(np.asarray(mix, np.float32))
    ref = m.mean(0)
    return (m - ref.mean()) / ref.std(), ref


def demucs_segments(mix, sd, cfg: HTConfig, shifts, offsets, overlap, k0, k1):
    m, _ = _standardized(mix)
    n = m.shape[1]
    plan, _, max_shift = segment_plan(n, cfg, shifts, offsets, overlap)
    padded = F.pad(m, (max_shift, max_shift))
    outs = []
    for (si, off, vl, o, clen) in plan[k0:k1]:
        x, _ = _padded(padded[None], off + o, clen, cfg.training_length)
        outs.append(ht_forward(x.numpy(), sd, cfg)[0])
    S = len(cfg.sources)
    return np.stack(outs).astype(np.float32) if outs else np.zeros((0, S, 2, cfg.training_length), np.float32)


def demucs_fold(mix, chunks, cfg: HTConfig, shifts, offsets, overlap):
    m, ref = _standardized(mix)
    n = m.shape[1]
    plan, stride, max_shift = segment_plan(n, cfg, shifts, offsets, overlap)
    seg = cfg.training_length
    S = len(cfg.sources)
    weight = torch.cat([torch.arange(1, seg // 2 + 1), torch.arange(seg - seg // 2, 0, -1)])
    weight = (weight / weight.max()) ** 1.0
    total = 0
    for si in range(max(shifts, 1)):
        items = [(k, p) for k, p in enumerate(plan) if p[0] == si]
        off, vl = items[0][1][1], items[0][1][2]
        out = torch.zeros(S, 2, vl)
        sw = torch.zeros(vl)
        for k, (_, _, _, o, clen) in items:
            y = _center_trim(torch.tensor(chunks[k]), clen)
            out[..., o:o + seg] += weight[:clen] * y
            sw[o:o + seg] += weight[:clen]
        out = out / sw
        total = total + out[..., max_shift - off:]
    src = total / max(shifts, 1)
    src = (src * ref.std() + ref.mean()).numpy()
    src[[0, 1]] = src[[1, 0]]
    return src
