"""CPU oracle for Demucs v3 (HDemucs, the `hdemucs_mmi` model of the reference's Demucs list).

TEST INFRASTRUCTURE ONLY (see oracle/mdx_oracle.py).  Restates ``uvr_lib_v5/demucs/hdemucs.py`` (HDemucs.__init__ :362-571
layer plan, _spec :573-597, _ispec :599-616, forward :670-782, HEncLayer :67-170, HDecLayer :252-330) and
``demucs.py`` (BLSTM :19-66, DConv :99-179, LocalState :152-221) for the default hybrid structure: CaC, no Wiener
filtering, no MultiWrap, dconv_mode = 1, hybrid_old = False.  nn.LSTM itself is torch's (the reference uses the same
module); everything around it is restated.

Parity status: PINNED on golden vectors written by the reference HDemucs class
(tests/golden/make_golden_hdemucs.py -> hdemucs_small.npz).  No HIP path is built on this oracle yet (next round).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

from .demucs_oracle import _pad1d_reflect, apply_model


@dataclass
class HDConfig:
    sources: tuple = ("drums", "bass", "other", "vocals")
    audio_channels: int = 2
    channels: int = 48
    growth: int = 2
    nfft: int = 4096
    depth: int = 6
    kernel_size: int = 8
    time_stride: int = 2
    stride: int = 4
    context: int = 1
    norm_starts: int = 4
    norm_groups: int = 4
    dconv_depth: int = 2
    dconv_comp: int = 4
    dconv_attn: int = 4
    dconv_lstm: int = 4
    freq_emb: float = 0.2
    samplerate: int = 44100
    segment: int = 40

    def ctor_kwargs(self) -> dict:
        return dict(sources=list(self.sources), audio_channels=self.audio_channels, channels=self.channels, growth=self.growth,
                    nfft=self.nfft, depth=self.depth, kernel_size=self.kernel_size, time_stride=self.time_stride, stride=self.stride,
                    context=self.context, norm_starts=self.norm_starts, norm_groups=self.norm_groups, dconv_depth=self.dconv_depth,
                    dconv_comp=self.dconv_comp, dconv_attn=self.dconv_attn, dconv_lstm=self.dconv_lstm, freq_emb=self.freq_emb,
                    samplerate=self.samplerate, segment=self.segment, rescale=0)

    def layers(self):
        """The per-layer plan of HDemucs.__init__ (:493-571)."""
        S = len(self.sources)
        out = []
        chin, chin_z = self.audio_channels, self.audio_channels * 2
        chout = chout_z = self.channels
        freqs = self.nfft // 2
        for index in range(self.depth):
            freq = freqs > 1
            ker, stri = self.kernel_size, self.stride
            if not freq:
                ker, stri = self.time_stride * 2, self.time_stride
            pad, last_freq = True, False
            if freq and freqs <= self.kernel_size:
                ker, pad, last_freq = freqs, False, True
            if last_freq:
                chout_z = max(chout, chout_z)
                chout = chout_z
            L = dict(index=index, freq=freq, ker=ker, stride=stri, pad=(ker // 4 if pad else 0), last_freq=last_freq,
                     norm=index >= self.norm_starts, lstm=index >= self.dconv_lstm, attn=index >= self.dconv_attn, chin_z=chin_z,
                     chout_z=chout_z, chin=chin, chout=chout, tenc=freq, freqs_in=freqs)
            if index == 0:
                chin = self.audio_channels * S
                chin_z = chin * 2
            L["dec_out_z"], L["dec_out"] = chin_z, chin
            out.append(L)
            chin, chin_z = chout, chout_z
            chout, chout_z = int(self.growth * chout), int(self.growth * chout_z)
            if freq:
                freqs = 1 if freqs <= self.kernel_size else freqs // self.stride
        return out


def make_hd_state(cfg: HDConfig, seed: int = 0) -> dict:
    """Seeded synthetic weights with HDemucs' state_dict names and shapes."""
    gen = torch.Generator().manual_seed(seed)
    sd: dict = {}

    def rn(*shape, scale=1.0):
        return torch.randn(*shape, generator=gen) * scale

    def gn(p, c):
        sd[p + ".weight"] = 0.8 + 0.4 * torch.rand(c, generator=gen)
        sd[p + ".bias"] = rn(c, scale=0.1)

    def conv(p, cout, cin, *k, scale=None):
        fan = cin * int(np.prod(k))
        sd[p + ".weight"] = rn(cout, cin, *k, scale=scale or math.sqrt(2.0 / fan))
        sd[p + ".bias"] = rn(cout, scale=0.05)

    def dconv(prefix, ch, lstm, attn):
        hidden = int(ch / cfg.dconv_comp)
        for d in range(cfg.dconv_depth):
            p = f"{prefix}.layers.{d}"
            i = 3
            conv(f"{p}.0", hidden, ch, 3, scale=math.sqrt(1.0 / (3 * ch)))
            gn(f"{p}.1", hidden)
            if lstm:
                for layer in range(2):
                    din = hidden if layer == 0 else 2 * hidden
                    for sfx in ("", "_reverse"):
                        sd[f"{p}.{i}.lstm.weight_ih_l{layer}{sfx}"] = rn(4 * hidden, din, scale=math.sqrt(1.0 / din))
                        sd[f"{p}.{i}.lstm.weight_hh_l{layer}{sfx}"] = rn(4 * hidden, hidden, scale=math.sqrt(1.0 / hidden))
                        sd[f"{p}.{i}.lstm.bias_ih_l{layer}{sfx}"] = rn(4 * hidden, scale=0.1)
                        sd[f"{p}.{i}.lstm.bias_hh_l{layer}{sfx}"] = rn(4 * hidden, scale=0.1)
                sd[f"{p}.{i}.linear.weight"] = rn(hidden, 2 * hidden, scale=math.sqrt(1.0 / (2 * hidden)))
                sd[f"{p}.{i}.linear.bias"] = rn(hidden, scale=0.05)
                i += 1
            if attn:
                for nm in ("content", "query", "key", "proj"):
                    conv(f"{p}.{i}.{nm}", hidden, hidden, 1, scale=math.sqrt(1.0 / hidden))
                conv(f"{p}.{i}.query_decay", 4 * 4, hidden, 1, scale=0.3 * math.sqrt(1.0 / hidden))
                sd[f"{p}.{i}.query_decay.bias"] = -2 + rn(16, scale=0.3)
                i += 1
            conv(f"{p}.{i}", 2 * ch, hidden, 1, scale=math.sqrt(1.0 / hidden))
            gn(f"{p}.{i + 1}", 2 * ch)
            sd[f"{p}.{i + 3}.scale"] = 0.3 + 0.4 * torch.rand(ch, generator=gen)

    Ls = cfg.layers()
    D = cfg.depth
    nt = sum(1 for L in Ls if L["tenc"])
    for L in Ls:
        i = L["index"]
        k2 = (L["ker"], 1) if L["freq"] else (L["ker"],)
        one = (1, 1) if L["freq"] else (1,)
        conv(f"encoder.{i}.conv", L["chout_z"], L["chin_z"], *k2)
        if L["norm"]:
            gn(f"encoder.{i}.norm1", L["chout_z"])
            gn(f"encoder.{i}.norm2", 2 * L["chout_z"])
        conv(f"encoder.{i}.rewrite", 2 * L["chout_z"], L["chout_z"], *one)
        dconv(f"encoder.{i}.dconv", L["chout_z"], L["lstm"], L["attn"])
        j = D - 1 - i
        c3 = (3, 3) if L["freq"] else (3,)
        sd[f"decoder.{j}.conv_tr.weight"] = rn(L["chout_z"], L["dec_out_z"], *k2, scale=math.sqrt(1.0 / L["chout_z"]))
        sd[f"decoder.{j}.conv_tr.bias"] = rn(L["dec_out_z"], scale=0.05)
        conv(f"decoder.{j}.rewrite", 2 * L["chout_z"], L["chout_z"], *c3)
        if L["norm"]:
            gn(f"decoder.{j}.norm2", L["dec_out_z"])
            gn(f"decoder.{j}.norm1", 2 * L["chout_z"])
        if L["tenc"]:
            K = cfg.kernel_size
            conv(f"tencoder.{i}.conv", L["chout"], L["chin"], K)
            jt = nt - 1 - i
            sd[f"tdecoder.{jt}.conv_tr.weight"] = rn(L["chout"], L["dec_out"], K, scale=math.sqrt(1.0 / L["chout"]))
            sd[f"tdecoder.{jt}.conv_tr.bias"] = rn(L["dec_out"], scale=0.05)
            if L["norm"]:
                gn(f"tdecoder.{jt}.norm2", L["dec_out"])
            if not L["last_freq"]:
                if L["norm"]:
                    gn(f"tencoder.{i}.norm1", L["chout"])
                    gn(f"tencoder.{i}.norm2", 2 * L["chout"])
                    gn(f"tdecoder.{jt}.norm1", 2 * L["chout"])
                conv(f"tencoder.{i}.rewrite", 2 * L["chout"], L["chout"], 1)
                dconv(f"tencoder.{i}.dconv", L["chout"], L["lstm"], L["attn"])
                conv(f"tdecoder.{jt}.rewrite", 2 * L["chout"], L["chout"], 3)
    sd["freq_emb.embedding.weight"] = rn(cfg.nfft // 2 // cfg.stride, cfg.channels, scale=0.05)
    return {k: v.float().contiguous() for k, v in sd.items()}


# --------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------
_LSTM_CACHE: dict = {}


def _blstm(x, sd, p):
    """BLSTM(dim, layers=2, max_steps=200, skip=True).forward (demucs.py:33-66): x [B, C, T]."""
    B, C, T = x.shape
    y = x
    key = (id(sd), p)
    if key not in _LSTM_CACHE:
        m = torch.nn.LSTM(bidirectional=True, num_layers=2, hidden_size=C, input_size=C)
        m.load_state_dict({k[len(p) + 6:]: v for k, v in sd.items() if k.startswith(p + ".lstm.")})
        _LSTM_CACHE[key] = m.eval()
    lstm = _LSTM_CACHE[key]
    framed = False
    width, stride = 200, 100
    if T > width:
        # utils.unfold (utils.py:36-50): frames of `width` with `stride`, zero padded to cover T
        n_frames = int(math.ceil(T / stride))
        tgt = (n_frames - 1) * stride + width
        xp = F.pad(x, (0, tgt - T))
        frames = xp.unfold(-1, width, stride)                      # B, C, nframes, width
        nframes = frames.shape[2]
        framed = True
        x = frames.permute(0, 2, 1, 3).reshape(-1, C, width)
    x = x.permute(2, 0, 1)
    x = lstm(x)[0]
    x = F.linear(x, sd[p + ".linear.weight"], sd[p + ".linear.bias"])
    x = x.permute(1, 2, 0)
    if framed:
        out = []
        frames = x.reshape(B, -1, C, width)
        limit = stride // 2
        for k in range(nframes):
            if k == 0:
                out.append(frames[:, k, :, :-limit])
            elif k == nframes - 1:
                out.append(frames[:, k, :, limit:])
            else:
                out.append(frames[:, k, :, limit:-limit])
        x = torch.cat(out, -1)[..., :T]
    return x + y


def _local_state(x, sd, p, heads=4, ndecay=4):
    """LocalState.forward (demucs.py:197-221), nfreqs = 0."""
    B, C, T = x.shape
    idx = torch.arange(T, dtype=x.dtype)
    delta = idx[:, None] - idx[None, :]
    c1 = lambda n: F.conv1d(x, sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"])  # noqa: E731
    q = c1("query").view(B, heads, -1, T)
    k = c1("key").view(B, heads, -1, T)
    dots = torch.einsum("bhct,bhcs->bhts", k, q) / k.shape[2] ** 0.5
    decays = torch.arange(1, ndecay + 1, dtype=x.dtype)
    dq = torch.sigmoid(c1("query_decay").view(B, heads, -1, T)) / 2
    dk = -decays.view(-1, 1, 1) * delta.abs() / ndecay ** 0.5
    dots = dots + torch.einsum("fts,bhfs->bhts", dk, dq)
    dots.masked_fill_(torch.eye(T, dtype=torch.bool), -100)
    w = torch.softmax(dots, dim=2)
    content = c1("content").view(B, heads, -1, T)
    res = torch.einsum("bhts,bhct->bhcs", w, content).reshape(B, -1, T)
    return x + F.conv1d(res, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def _dconv(x, sd, prefix, cfg: HDConfig, lstm, attn):
    for d in range(cfg.dconv_depth):
        p = f"{prefix}.layers.{d}"
        dil = 2 ** d
        i = 3
        y = F.conv1d(x, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], dilation=dil, padding=dil)
        y = F.gelu(F.group_norm(y, 1, sd[f"{p}.1.weight"], sd[f"{p}.1.bias"]))
        if lstm:
            y = _blstm(y, sd, f"{p}.{i}")
            i += 1
        if attn:
            y = _local_state(y, sd, f"{p}.{i}")
            i += 1
        y = F.conv1d(y, sd[f"{p}.{i}.weight"], sd[f"{p}.{i}.bias"])
        y = F.glu(F.group_norm(y, 1, sd[f"{p}.{i + 1}.weight"], sd[f"{p}.{i + 1}.bias"]), dim=1)
        x = x + sd[f"{p}.{i + 3}.scale"][:, None] * y
    return x


def _norm(y, sd, name, cfg, on):
    return F.group_norm(y, cfg.norm_groups, sd[name + ".weight"], sd[name + ".bias"]) if on else y


def _enc(x, sd, p, L, cfg, freq, empty=False, inject=None):
    """HEncLayer.forward (hdemucs.py:139-170)."""
    if not freq and x.dim() == 4:
        B, C, Fr, T = x.shape
        x = x.view(B, -1, T)
    ker = L["ker"] if (freq or not L["freq"]) else cfg.kernel_size
    stride = L["stride"] if (freq or not L["freq"]) else cfg.stride
    pad = L["pad"] if (freq or not L["freq"]) else cfg.kernel_size // 4
    if not freq:
        le = x.shape[-1]
        if le % stride:
            x = F.pad(x, (0, stride - (le % stride)))
        y = F.conv1d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=stride, padding=pad)
    else:
        y = F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=(stride, 1), padding=(pad, 0))
    if empty:
        return y
    if inject is not None:
        if inject.dim() == 3 and y.dim() == 4:
            inject = inject[:, :, None]
        y = y + inject
    y = F.gelu(_norm(y, sd, p + ".norm1", cfg, L["norm"]))
    if freq:
        B, C, Fr, T = y.shape
        y = _dconv(y.permute(0, 2, 1, 3).reshape(-1, C, T), sd, p + ".dconv", cfg, L["lstm"], L["attn"])
        y = y.view(B, Fr, C, T).permute(0, 2, 1, 3)
        z = F.conv2d(y, sd[p + ".rewrite.weight"], sd[p + ".rewrite.bias"])
    else:
        y = _dconv(y, sd, p + ".dconv", cfg, L["lstm"], L["attn"])
        z = F.conv1d(y, sd[p + ".rewrite.weight"], sd[p + ".rewrite.bias"])
    return F.glu(_norm(z, sd, p + ".norm2", cfg, L["norm"]), dim=1)


def _dec(x, skip, length, sd, p, L, cfg, freq, last, empty=False):
    """HDecLayer.forward (hdemucs.py:303-330) -> (z, pre)."""
    tb = not freq and L["freq"]                      # a time-branch layer paired with a frequency layer
    ker = cfg.kernel_size if tb else L["ker"]
    stride = cfg.stride if tb else L["stride"]
    pad = cfg.kernel_size // 4 if tb else L["pad"]
    if freq and x.dim() == 3:
        B, C, T = x.shape
        x = x.view(B, L["chout_z"], -1, T)
    if not empty:
        x = x + skip
        if freq:
            y = F.conv2d(x, sd[p + ".rewrite.weight"], sd[p + ".rewrite.bias"], padding=1)
        else:
            y = F.conv1d(x, sd[p + ".rewrite.weight"], sd[p + ".rewrite.bias"], padding=1)
        y = F.glu(_norm(y, sd, p + ".norm1", cfg, L["norm"]), dim=1)
    else:
        y = x
    if freq:
        z = F.conv_transpose2d(y, sd[p + ".conv_tr.weight"], sd[p + ".conv_tr.bias"], stride=(stride, 1))
    else:
        z = F.conv_transpose1d(y, sd[p + ".conv_tr.weight"], sd[p + ".conv_tr.bias"], stride=stride)
    z = _norm(z, sd, p + ".norm2", cfg, L["norm"])
    if freq:
        if pad:
            z = z[..., pad:-pad, :]
    else:
        z = z[..., pad:pad + length]
    return (z if last else F.gelu(z)), y


@torch.no_grad()
def hd_forward(mix, sd: dict, cfg: HDConfig, taps: dict | None = None):
    """HDemucs.forward (hdemucs.py:670-782), eval: [B, 2, L] -> [B, S, 2, L].  taps: filled with intermediates (debugging)."""
    mix = torch.as_tensor(np.ascontiguousarray(mix), dtype=torch.float32)
    length = mix.shape[-1]
    hl = cfg.nfft // 4
    le = int(math.ceil(length / hl))
    pad = hl // 2 * 3
    xp = _pad1d_reflect(mix, pad, pad + le * hl - length)
    B, C, Lp = xp.shape
    z = torch.stft(xp.reshape(-1, Lp), cfg.nfft, hl, window=torch.hann_window(cfg.nfft), win_length=cfg.nfft, normalized=True,
                   center=True, return_complex=True, pad_mode="reflect")
    z = z.view(B, C, z.shape[-2], z.shape[-1])[..., :-1, :][..., 2:2 + le]
    Fq, T = z.shape[-2], z.shape[-1]
    x = torch.view_as_real(z).permute(0, 1, 4, 2, 3).reshape(B, C * 2, Fq, T)
    mean, std = x.mean(dim=(1, 2, 3), keepdim=True), x.std(dim=(1, 2, 3), keepdim=True)
    x = (x - mean) / (1e-5 + std)
    xt = mix
    meant, stdt = xt.mean(dim=(1, 2), keepdim=True), xt.std(dim=(1, 2), keepdim=True)
    xt = (xt - meant) / (1e-5 + stdt)
    Ls = cfg.layers()
    nt = sum(1 for L in Ls if L["tenc"])
    saved, saved_t, lengths, lengths_t = [], [], [], []
    for L in Ls:
        i = L["index"]
        lengths.append(x.shape[-1])
        inject = None
        if i < nt:
            lengths_t.append(xt.shape[-1])
            xt = _enc(xt, sd, f"tencoder.{i}", L, cfg, freq=False, empty=L["last_freq"])
            if not L["last_freq"]:
                saved_t.append(xt)
                if taps is not None:
                    taps[f"skt{i}"] = xt
            else:
                inject = xt
                if taps is not None:
                    taps["inj"] = xt
        x = _enc(x, sd, f"encoder.{i}", L, cfg, freq=L["freq"], inject=inject)
        if i == 0 and cfg.freq_emb:
            frs = torch.arange(x.shape[-2])
            emb = (F.embedding(frs, sd["freq_emb.embedding.weight"]) * 10.0).t()[None, :, :, None].expand_as(x)
            x = x + cfg.freq_emb * emb
        saved.append(x)
        if taps is not None:
            taps[f"skf{i}"] = x
    x = torch.zeros_like(x)
    xt = torch.zeros_like(x)
    offset = cfg.depth - nt
    for j in range(cfg.depth):
        L = Ls[cfg.depth - 1 - j]
        skip = saved.pop(-1)
        x, pre = _dec(x, skip, lengths.pop(-1), sd, f"decoder.{j}", L, cfg, freq=L["freq"], last=L["index"] == 0)
        if taps is not None:
            taps[f"dec{j}"] = x
            taps[f"pre{j}"] = pre
        if j >= offset:
            jt = j - offset
            Lt = Ls[nt - 1 - jt]
            length_t = lengths_t.pop(-1)
            if Lt["last_freq"]:
                xt, _ = _dec(pre[:, :, 0], None, length_t, sd, f"tdecoder.{jt}", Lt, cfg, freq=False, last=Lt["index"] == 0, empty=True)
            else:
                xt, _ = _dec(xt, saved_t.pop(-1), length_t, sd, f"tdecoder.{jt}", Lt, cfg, freq=False, last=Lt["index"] == 0)
            if taps is not None:
                taps[f"tdec{jt}"] = xt
    S = len(cfg.sources)
    x = x.view(B, S, -1, Fq, T) * std[:, None] + mean[:, None]
    zout = torch.view_as_complex(x.view(B, S, -1, 2, Fq, T).permute(0, 1, 2, 4, 5, 3).contiguous())
    zp = F.pad(F.pad(zout, (0, 0, 0, 1)), (2, 2))
    lei = hl * int(math.ceil(length / hl)) + 2 * pad
    shp = zp.shape
    xo = torch.istft(zp.reshape(-1, shp[-2], shp[-1]), cfg.nfft, hl, window=torch.hann_window(cfg.nfft), win_length=cfg.nfft,
                     normalized=True, length=lei, center=True)
    xo = xo.view(*shp[:-2], lei)[..., pad:pad + length]
    xt = xt.view(B, S, -1, length) * stdt[:, None] + meant[:, None]
    return (xt + xo).numpy()


def demix_hdemucs(mix: np.ndarray, sd: dict, cfg: HDConfig, shifts=2, overlap=0.25, split=True, offsets=None) -> np.ndarray:
    """DemucsSeparator.demix_demucs (demucs_separator.py:162-194) around HDemucs: [2, N] -> [S, 2, N] (stems 0/1 swapped).
    apply_model's leaf call runs each chunk at its own length (HDemucs has no valid_length, apply.py:251-256)."""
    m = torch.tensor(np.asarray(mix, np.float32))
    ref = m.mean(0)
    m = (m - ref.mean()) / ref.std()
    fn = lambda x: hd_forward(x.numpy() if hasattr(x, "numpy") else x, sd, cfg)  # noqa: E731
    src = apply_model(fn, m[None], cfg, shifts=shifts, split=split, overlap=overlap, offsets=offsets)[0]
    src = (src * ref.std() + ref.mean()).numpy()
    src[[0, 1]] = src[[1, 0]]
    return src


# --------------------------------------------------------------------------
# chunk-list form of demix_hdemucs (test double of asx_hd_plan / asx_hd_segments_dev / asx_hd_fold_dev): every chunk runs at
# its own length; row k of the slab holds it from column 0
# --------------------------------------------------------------------------
def hd_segment_plan(n: int, cfg: HDConfig, shifts, offsets, overlap=0.25):
    seg = int(cfg.samplerate * cfg.segment)
    stride = int((1 - overlap) * seg)
    max_shift = int(0.5 * cfg.samplerate) if shifts else 0
    plan = []
    for si in range(max(shifts, 1)):
        off = offsets[si] if shifts else 0
        vl = n + max_shift - off
        plan += [(si, off, vl, o, min(vl - o, seg)) for o in range(0, vl, stride)]
    return plan, stride, max_shift, seg


def _std(mix):
    m = torch.tensor(np.asarray(mix, np.float32))
    ref = m.mean(0)
    return (m - ref.mean()) / ref.std(), ref


def hd_segments(mix, sd, cfg: HDConfig, shifts, offsets, overlap, k0, k1):
    m, _ = _std(mix)
    plan, _, max_shift, seg = hd_segment_plan(m.shape[1], cfg, shifts, offsets, overlap)
    padded = F.pad(m, (max_shift, max_shift))
    out = np.zeros((k1 - k0, len(cfg.sources), 2, seg), np.float32)
    for i, (si, off, vl, o, clen) in enumerate(plan[k0:k1]):
        out[i, :, :, :clen] = hd_forward(padded[None, :, off + o: off + o + clen].numpy(), sd, cfg)[0]
    return out


def hd_fold(mix, chunks, cfg: HDConfig, shifts, offsets, overlap):
    m, ref = _std(mix)
    plan, stride, max_shift, seg = hd_segment_plan(m.shape[1], cfg, shifts, offsets, overlap)
    weight = torch.cat([torch.arange(1, seg // 2 + 1), torch.arange(seg - seg // 2, 0, -1)])
    weight = weight / weight.max()
    total = 0
    for si in range(max(shifts, 1)):
        items = [(k, p) for k, p in enumerate(plan) if p[0] == si]
        off, vl = items[0][1][1], items[0][1][2]
        out = torch.zeros(len(cfg.sources), 2, vl)
        sw = torch.zeros(vl)
        for k, (_, _, _, o, clen) in items:
            out[..., o:o + seg] += weight[:clen] * torch.tensor(chunks[k][..., :clen])
            sw[o:o + seg] += weight[:clen]
        total = total + (out / sw)[..., max_shift - off:]
    src = total / max(shifts, 1)
    src = (src * ref.std() + ref.mean()).numpy()
    src[[0, 1]] = src[[1, 0]]
    return src
