"""CPU oracle for the BS-Roformer demix path.

TEST INFRASTRUCTURE ONLY (see oracle/mdx_oracle.py for the rules).  Restates
``uvr_lib_v5/roformer/bs_roformer.py`` (BSRoformer.forward :418-522 with RMSNorm :42,
FeedForward :55, Attention :68, Transformer :136, BandSplit :163, MaskEstimator :205),
``attend.py`` (softmax attention :66-112) and the Roformer branch of
``MDXCSeparator.demix`` (architectures/mdxc_separator.py:272-343) in torch-CPU fp32.

Third-party piece that is absent here: ``rotary_embedding_torch`` 0.6.5
(``RotaryEmbedding(dim).rotate_queries_or_keys``).  Its published algorithm is restated in
``RotaryEmbedding`` below (theta = 10000, freqs = theta^(-arange(0, d, 2)/d), positions
arange(n), interleaved pairs, x*cos + rotate_half(x)*sin); the golden generator installs this
restatement as the reference's dependency, so parity of the rotary step is pinned on the
restatement, not on the third-party package -- flagged in DESIGN.md.

Parity status otherwise: PINNED on golden vectors written by the reference BSRoformer /
MDXCSeparator classes (tests/golden/make_golden_roformer.py -> roformer_small.npz).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import scipy.signal
import torch
import torch.nn.functional as F


class RotaryEmbedding(torch.nn.Module):
    """Restatement of rotary_embedding_torch.RotaryEmbedding (freqs_for='lang', defaults)."""

    def __init__(self, dim, theta=10000):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        self.freqs = torch.nn.Parameter(freqs, requires_grad=False)

    def rotate_queries_or_keys(self, t, seq_dim=-2):
        n = t.shape[seq_dim]
        pos = torch.arange(n, dtype=t.dtype, device=t.device)
        fr = torch.einsum("i,j->ij", pos, self.freqs.to(t.dtype))
        fr = fr.repeat_interleave(2, dim=-1)                      # '... n -> ... (n r)', r = 2
        return rope_apply(t, fr)


def rope_apply(t, fr):
    x = t.reshape(*t.shape[:-1], -1, 2)
    x1, x2 = x.unbind(-1)
    rot = torch.stack((-x2, x1), dim=-1).reshape(t.shape)         # rotate_half
    return t * fr.cos() + rot * fr.sin()


@dataclass
class RoformerConfig:
    """config.model / config.audio fields BSRoformer reads (roformer_loader.py:123-150)."""
    dim: int = 512
    depth: int = 12
    stereo: bool = True
    num_stems: int = 1
    time_transformer_depth: int = 1
    freq_transformer_depth: int = 1
    freqs_per_bands: tuple = ()
    dim_head: int = 64
    heads: int = 8
    mlp_expansion_factor: int = 4
    mask_estimator_depth: int = 2
    stft_n_fft: int = 2048
    stft_hop_length: int = 441
    stft_win_length: int = 2048
    dim_t: int = 801
    sample_rate: int = 44100
    instruments: tuple = ("vocals", "other")
    target_instrument: str | None = "vocals"
    stft_normalized: bool = False        # torch.stft / istft normalized=True (bs_roformer.py:332, 384)
    stft_window_fn: str = "hann_window"  # name of the torch window function (bs_roformer.py:333, 386: default torch.hann_window)
    mel: bool = False            # MelBandRoformer (mel_band_roformer.py): overlapping mel bands, see mel_band_layout
    num_bands: int = 60
    band_starts: tuple = ()      # mel only: first frequency bin of each band

    @staticmethod
    def mel_config(**kw) -> "RoformerConfig":
        c = RoformerConfig(mel=True, mask_estimator_depth=kw.pop("mask_estimator_depth", 1), **kw)
        starts, counts = mel_band_layout(c.sample_rate, c.stft_n_fft, c.num_bands)
        c.freqs_per_bands = tuple(counts)
        c.band_starts = tuple(starts)
        return c

    @property
    def audio_channels(self):
        return 2 if self.stereo else 1

    @property
    def band_dims(self):
        return tuple(2 * f * self.audio_channels for f in self.freqs_per_bands)

    def model_kwargs(self) -> dict:
        if self.mel:
            return dict(dim=self.dim, depth=self.depth, stereo=self.stereo, num_stems=self.num_stems,
                        time_transformer_depth=self.time_transformer_depth, freq_transformer_depth=self.freq_transformer_depth,
                        num_bands=self.num_bands, dim_head=self.dim_head, heads=self.heads,
                        mlp_expansion_factor=self.mlp_expansion_factor, dim_freqs_in=self.stft_n_fft // 2 + 1,
                        sample_rate=self.sample_rate, stft_n_fft=self.stft_n_fft, stft_hop_length=self.stft_hop_length,
                        stft_win_length=self.stft_win_length, mask_estimator_depth=self.mask_estimator_depth)
        kw = dict(dim=self.dim, depth=self.depth, stereo=self.stereo, num_stems=self.num_stems,
                  time_transformer_depth=self.time_transformer_depth,
                  freq_transformer_depth=self.freq_transformer_depth, freqs_per_bands=tuple(self.freqs_per_bands),
                  dim_head=self.dim_head, heads=self.heads, mlp_expansion_factor=self.mlp_expansion_factor,
                  stft_n_fft=self.stft_n_fft, stft_hop_length=self.stft_hop_length,
                  stft_win_length=self.stft_win_length)
        if self.stft_normalized:
            kw["stft_normalized"] = True
        if self.stft_window_fn != "hann_window":
            kw["stft_window_fn"] = getattr(torch, self.stft_window_fn)
        return kw

    def as_model_data(self) -> dict:
        m = self.model_kwargs()
        if not self.mel:
            m["freqs_per_bands"] = list(self.freqs_per_bands)
        if "stft_window_fn" in m:                      # a YAML carries a callable as its dotted name
            m["stft_window_fn"] = "torch." + self.stft_window_fn
        return {"audio": {"sample_rate": self.sample_rate, "hop_length": self.stft_hop_length, "n_fft": self.stft_n_fft,
                          "num_channels": 2, "dim_f": self.stft_n_fft // 2, "chunk_size": self.stft_hop_length * (self.dim_t - 1)},
                "model": m, "training": {"instruments": list(self.instruments), "target_instrument": self.target_instrument},
                "inference": {"dim_t": self.dim_t}, "is_roformer": True}


def mel_band_layout(sr: int, n_fft: int, n_mels: int):
    """(first bin, bin count) of every band of MelBandRoformer (mel_band_roformer.py:279-300): the support of
    librosa.filters.mel(sr, n_fft, n_mels) (librosa absent -- restated from its published definition: Slaney mel scale,
    triangular filters between consecutive mel points, `> 0` pattern only; PARITY UNPINNED against librosa itself), with
    bin 0 forced into the first band and the last bin into the last.  Raises if a band is not one contiguous run."""
    def hz_to_mel(f):
        f = np.asarray(f, np.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        m = np.asarray(m, np.float64)
        f_sp = 200.0 / 3
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(float(sr) / 2), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, len(fftfreqs)), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis].astype(np.float32)
    weights[0][0] = 1.0
    weights[-1, -1] = 1.0
    pattern = weights > 0
    if not pattern.any(axis=0).all():
        raise ValueError("all frequencies need to be covered by all bands for now")
    starts, counts = [], []
    for i in range(n_mels):
        idx = np.nonzero(pattern[i])[0]
        if len(idx) == 0 or idx[-1] - idx[0] + 1 != len(idx):
            raise ValueError(f"mel band {i} is empty or not contiguous")
        starts.append(int(idx[0]))
        counts.append(int(len(idx)))
    return starts, counts


def mel_filter_stub(sr, n_fft, n_mels):
    """stand-in for librosa.filters.mel used by the golden script: a float32 matrix with the same `> 0` support"""
    starts, counts = mel_band_layout(sr, n_fft, n_mels)
    w = np.zeros((n_mels, n_fft // 2 + 1), np.float32)
    for i, (s0, c) in enumerate(zip(starts, counts)):
        w[i, s0:s0 + c] = 1.0
    # the reference itself forces [0][0] and [-1, -1]; leave them as the true filterbank would have them
    return w


DEFAULT_FREQS_PER_BANDS = (2,) * 24 + (4,) * 12 + (12,) * 8 + (24,) * 8 + (48,) * 8 + (128, 129)


def make_roformer_state(cfg: RoformerConfig, seed: int = 0) -> dict:
    """Seeded synthetic weights with BSRoformer's state_dict names and shapes."""
    gen = torch.Generator().manual_seed(seed)
    sd: dict = {}
    d, inner = cfg.dim, cfg.heads * cfg.dim_head

    def lin(name, n_out, n_in, bias=True, scale=1.0):
        sd[name + ".weight"] = torch.randn(n_out, n_in, generator=gen) * (scale / math.sqrt(n_in))
        if bias:
            sd[name + ".bias"] = 0.05 * torch.randn(n_out, generator=gen)

    def gamma(name, n):
        sd[name] = 0.8 + 0.4 * torch.rand(n, generator=gen)

    rot = RotaryEmbedding(cfg.dim_head).freqs.detach().clone()
    for i in range(cfg.depth):
        for k, tdepth in enumerate((cfg.time_transformer_depth, cfg.freq_transformer_depth)):
            for j in range(tdepth):
                p = f"layers.{i}.{k}.layers.{j}"
                sd[f"{p}.0.rotary_embed.freqs"] = rot.clone()
                gamma(f"{p}.0.norm.gamma", d)
                lin(f"{p}.0.to_qkv", 3 * inner, d, bias=False)
                lin(f"{p}.0.to_gates", cfg.heads, d)
                lin(f"{p}.0.to_out.0", d, inner, bias=False, scale=0.5)
                gamma(f"{p}.1.net.0.gamma", d)
                lin(f"{p}.1.net.1", d * 4, d)
                lin(f"{p}.1.net.4", d, d * 4, scale=0.5)
    if cfg.mel:
        for i in range(cfg.depth):
            for k in range(2):
                gamma(f"layers.{i}.{k}.norm.gamma", d)     # Transformer(norm_output=True), mel_band_roformer.py:111
    else:
        gamma("final_norm.gamma", d)
    for j, din in enumerate(cfg.band_dims):
        gamma(f"band_split.to_features.{j}.0.gamma", din)
        lin(f"band_split.to_features.{j}.1", d, din)
    # MelBandRoformer builds its MaskEstimator without passing mlp_expansion_factor (mel_band_roformer.py:312): always 4
    hid = d * (4 if cfg.mel else cfg.mlp_expansion_factor)
    for s in range(cfg.num_stems):
        for j, din in enumerate(cfg.band_dims):
            p = f"mask_estimators.{s}.to_freqs.{j}.0"
            dims = (d,) + (hid,) * (cfg.mask_estimator_depth - (0 if cfg.mel else 1)) + (din * 2,)
            for li, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
                lin(f"{p}.{2 * li}", b, a)
    return {k: v.float().contiguous() for k, v in sd.items()}


def _rms(x, g):
    return F.normalize(x, dim=-1) * (x.shape[-1] ** 0.5) * g


def _attention(x, sd, p, cfg: RoformerConfig):
    h, dh = cfg.heads, cfg.dim_head
    xn = _rms(x, sd[p + ".norm.gamma"])
    qkv = F.linear(xn, sd[p + ".to_qkv.weight"])
    b, n, _ = qkv.shape
    q, k, v = qkv.reshape(b, n, 3, h, dh).permute(2, 0, 3, 1, 4)          # qkv b h n d
    pos = torch.arange(n, dtype=x.dtype)
    fr = torch.einsum("i,j->ij", pos, sd[p + ".rotary_embed.freqs"]).repeat_interleave(2, dim=-1)
    q, k = rope_apply(q, fr), rope_apply(k, fr)
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * (dh ** -0.5)
    out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
    gates = F.linear(xn, sd[p + ".to_gates.weight"], sd[p + ".to_gates.bias"])    # b n h
    out = out * gates.permute(0, 2, 1).unsqueeze(-1).sigmoid()
    out = out.permute(0, 2, 1, 3).reshape(b, n, h * dh)
    return F.linear(out, sd[p + ".to_out.0.weight"])


def _ff(x, sd, p):
    y = _rms(x, sd[p + ".net.0.gamma"])
    y = F.gelu(F.linear(y, sd[p + ".net.1.weight"], sd[p + ".net.1.bias"]))
    return F.linear(y, sd[p + ".net.4.weight"], sd[p + ".net.4.bias"])


def _transformer(x, sd, prefix, depth, cfg):
    for j in range(depth):
        x = _attention(x, sd, f"{prefix}.layers.{j}.0", cfg) + x
        x = _ff(x, sd, f"{prefix}.layers.{j}.1") + x
    if cfg.mel:
        return _rms(x, sd[f"{prefix}.norm.gamma"])                         # norm_output=True (mel_band_roformer.py:111,120)
    return x                                                               # norm_output=False (bs_roformer.py:362)


@torch.no_grad()
def roformer_forward(wave, sd: dict, cfg: RoformerConfig):
    """BSRoformer.forward (bs_roformer.py:418-522), inference branch: [B,2,t] -> [B,(n,)2,t']."""
    raw = torch.as_tensor(np.ascontiguousarray(wave), dtype=torch.float32)
    b, s, t = raw.shape
    win = getattr(torch, cfg.stft_window_fn)(cfg.stft_win_length)
    st = torch.stft(raw.reshape(b * s, t), n_fft=cfg.stft_n_fft, hop_length=cfg.stft_hop_length,
                    win_length=cfg.stft_win_length, window=win, normalized=cfg.stft_normalized, return_complex=True)
    st = torch.view_as_real(st).reshape(b, s, st.shape[1], st.shape[2], 2)          # b s f t c
    stft_repr = st.permute(0, 2, 1, 3, 4).reshape(b, -1, st.shape[3], 2)           # b (f s) t c
    x = stft_repr.permute(0, 2, 1, 3).reshape(b, stft_repr.shape[2], -1)            # b t (f c)
    outs, off = [], 0
    per_bin = 2 * s
    for j, din in enumerate(cfg.band_dims):
        if cfg.mel:
            off = cfg.band_starts[j] * per_bin                                      # gather x[freq_indices] (:368-372)
        xb = _rms(x[..., off:off + din], sd[f"band_split.to_features.{j}.0.gamma"])
        outs.append(F.linear(xb, sd[f"band_split.to_features.{j}.1.weight"], sd[f"band_split.to_features.{j}.1.bias"]))
        off += din
    x = torch.stack(outs, dim=-2)                                                   # b t f d
    for i in range(cfg.depth):
        bb, tt, ff, dd = x.shape
        xt = x.permute(0, 2, 1, 3).reshape(bb * ff, tt, dd)
        xt = _transformer(xt, sd, f"layers.{i}.0", cfg.time_transformer_depth, cfg)
        x = xt.reshape(bb, ff, tt, dd).permute(0, 2, 1, 3)
        xf = x.reshape(bb * tt, ff, dd)
        xf = _transformer(xf, sd, f"layers.{i}.1", cfg.freq_transformer_depth, cfg)
        x = xf.reshape(bb, tt, ff, dd)
    if not cfg.mel:
        x = _rms(x, sd["final_norm.gamma"])
    masks = []
    for sidx in range(cfg.num_stems):
        outs = []
        for j, din in enumerate(cfg.band_dims):
            p = f"mask_estimators.{sidx}.to_freqs.{j}.0"
            y = x[:, :, j]
            nl = cfg.mask_estimator_depth + (1 if cfg.mel else 0)
            for li in range(nl):
                y = F.linear(y, sd[f"{p}.{2 * li}.weight"], sd[f"{p}.{2 * li}.bias"])
                if li < nl - 1:
                    y = torch.tanh(y)
            outs.append(F.glu(y, dim=-1))
        masks.append(torch.cat(outs, dim=-1))
    mask = torch.stack(masks, dim=1)                                                # b n t (f c)
    if cfg.mel:
        # scatter_add of the band masks onto their bins, averaged by the number of covering bands (:404-416)
        nbin = stft_repr.shape[1]                                                   # (f s)
        summed = torch.zeros(b, cfg.num_stems, mask.shape[2], nbin * 2)
        count = torch.zeros(nbin)
        off = 0
        for j, din in enumerate(cfg.band_dims):
            lo = cfg.band_starts[j] * s
            summed[..., lo * 2: lo * 2 + din] += mask[..., off:off + din]
            count[lo: lo + din // 2] += 1
            off += din
        mask = summed / count.clamp(min=1e-8).repeat_interleave(2)
    mask = mask.reshape(b, cfg.num_stems, mask.shape[2], -1, 2).permute(0, 1, 3, 2, 4)   # b n f t c
    z = torch.view_as_complex(stft_repr.unsqueeze(1).contiguous()) * torch.view_as_complex(mask.contiguous())
    z = z.reshape(b, cfg.num_stems, -1, s, z.shape[-1]).permute(0, 1, 3, 2, 4)      # b n s f t
    rec = torch.istft(z.reshape(b * cfg.num_stems * s, z.shape[3], z.shape[4]), n_fft=cfg.stft_n_fft,
                      hop_length=cfg.stft_hop_length, win_length=cfg.stft_win_length, window=win, normalized=cfg.stft_normalized,
                      return_complex=False)
    rec = rec.reshape(b, cfg.num_stems, s, -1)
    if cfg.num_stems == 1:
        rec = rec[:, 0]
    return rec.numpy()


def roformer_plan(n: int, cfg: RoformerConfig, overlap, segment_size=None):
    """mdxc_separator.py:276-306: (chunk_size, step, starts) -- starts already re-anchored at the tail."""
    seg = segment_size if segment_size is not None else cfg.dim_t
    chunk_size = cfg.stft_hop_length * (seg - 1)
    desired = int(overlap * cfg.sample_rate)
    step = chunk_size if desired <= 0 else min(desired, chunk_size)
    starts = []
    for i in range(0, n, step):
        starts.append(n - chunk_size if i + chunk_size > n else i)
    return chunk_size, step, starts


def roformer_demix(mix: np.ndarray, sd: dict, cfg: RoformerConfig, overlap=8, segment_size=None) -> np.ndarray:
    """Roformer branch of MDXCSeparator.demix (mdxc_separator.py:272-343): [2,N] -> [len(instruments),2,N]."""
    mix_t = torch.tensor(np.asarray(mix, np.float32))
    n = mix_t.shape[1]
    chunk_size, step, starts = roformer_plan(n, cfg, overlap, segment_size)
    if n < chunk_size:
        raise ValueError("mix shorter than one chunk")
    window = torch.tensor(scipy.signal.windows.hamming(chunk_size), dtype=torch.float32)
    req = (len(cfg.instruments), 2, n)
    result = torch.zeros(req)
    counter = torch.zeros(req)
    for st in starts:
        part = mix_t[:, st:st + chunk_size]
        x = torch.tensor(roformer_forward(part[None].numpy(), sd, cfg)[0])
        safe = min(chunk_size, x.shape[-1], window.shape[0])
        result[..., st:st + safe] += x[..., :safe] * window[:safe]
        counter[..., st:st + safe] += window[:safe]
    return (result / counter.clamp(min=1e-10)).numpy()


def roformer_chunks(mix: np.ndarray, sd: dict, cfg: RoformerConfig, overlap, k0: int, k1: int) -> np.ndarray:
    """model outputs of chunks [k0, k1) of roformer_demix's loop: [k1-k0, S, 2, chunk] (test double of asx_rof_chunks_dev)"""
    n = mix.shape[1]
    chunk_size, step, starts = roformer_plan(n, cfg, overlap)
    outs = []
    for st in starts[k0:k1]:
        x = roformer_forward(np.asarray(mix[None, :, st:st + chunk_size], np.float32), sd, cfg)[0]
        outs.append(x[None] if x.ndim == 2 else x)
    return np.stack(outs).astype(np.float32) if outs else np.zeros((0, cfg.num_stems, 2, chunk_size), np.float32)


def roformer_fold(chunks: np.ndarray, n: int, cfg: RoformerConfig, overlap) -> np.ndarray:
    """the Hamming fold of roformer_demix over precomputed chunk outputs (test double of asx_rof_finalize_dev)"""
    chunk_size, step, starts = roformer_plan(n, cfg, overlap)
    window = torch.tensor(scipy.signal.windows.hamming(chunk_size), dtype=torch.float32)
    req = (len(cfg.instruments), 2, n)
    result = torch.zeros(req)
    counter = torch.zeros(req)
    for k, st in enumerate(starts):
        x = torch.tensor(chunks[k])
        if cfg.num_stems == 1:
            x = x[0]
        safe = min(chunk_size, x.shape[-1], window.shape[0])
        result[..., st:st + safe] += x[..., :safe] * window[:safe]
        counter[..., st:st + safe] += window[:safe]
    return (result / counter.clamp(min=1e-10)).numpy()
