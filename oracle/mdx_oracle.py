"""CPU oracle for the chunked-spectrogram demix path (MDX / ConvTDFNet).

TEST INFRASTRUCTURE ONLY.  This module is a plain numpy / torch-CPU fp32
restatement of the reference algorithm.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker -- the product path (the HIP engine behind
``include/asx.h``) never calls into this file and fails loudly when the HIP
library is missing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
package itself (``/root/reference``; runnable only in the build container) and
writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every
function below against those vectors.  The one piece of the reference path that
is *not* available (onnxruntime 1.23.0 executing ``UVR-MDX-NET-Inst_HQ_3.onnx``,
call site mdx_separator.py:122-123) is restated from the reference's own torch
definition of that graph, ``uvr_lib_v5/mdxnet.py:30-120`` +
``uvr_lib_v5/modules.py:5-74``, and pinned against that class.

All ``file:line`` citations are relative to
``/root/reference/audio_separator/separator/``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import scipy.fft
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# STFT / iSTFT   (uvr_lib_v5/stft.py)
# --------------------------------------------------------------------------

def hann_periodic(n: int) -> np.ndarray:
    """torch.hann_window(n, periodic=True) (stft.py:18) as float32."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(np.float32)


def stft_forward(x: np.ndarray, n_fft: int, hop: int, dim_f: int) -> np.ndarray:
    """STFT.__call__ (stft.py:20-56).

    x: float32 [B, 2, C].  Returns float32 [B, 4, dim_f, T] with channel order
    (L_re, L_im, R_re, R_im), T = C // hop + 1.  torch.stft(center=True)
    reflect-pads n_fft/2 on both sides of every row (stft.py:41).
    """
    x = np.asarray(x, dtype=np.float32)
    B, ch, C = x.shape
    half = n_fft // 2
    xp = np.pad(x.reshape(B * ch, C), ((0, 0), (half, half)), mode="reflect")
    T = C // hop + 1
    idx = (np.arange(T) * hop)[:, None] + np.arange(n_fft)[None, :]
    frames = xp[:, idx] * hann_periodic(n_fft)[None, None, :]          # [B*ch, T, n]
    Z = scipy.fft.rfft(frames.astype(np.float32), axis=-1)             # complex64
    Z = Z[:, :, :dim_f]                                                 # crop (stft.py:56)
    out = np.empty((B * ch, 2, dim_f, T), dtype=np.float32)
    out[:, 0] = Z.real.transpose(0, 2, 1)
    out[:, 1] = Z.imag.transpose(0, 2, 1)
    return out.reshape(B, ch * 2, dim_f, T)


def istft_envelope(n_fft: int, hop: int, T: int) -> np.ndarray:
    """Sum of squared windows as torch.istft builds it (overlap-add of w^2)."""
    w = hann_periodic(n_fft).astype(np.float32)
    env = np.zeros(n_fft + hop * (T - 1), dtype=np.float32)
    w2 = (w * w).astype(np.float32)
    for t in range(T):
        env[t * hop: t * hop + n_fft] += w2
    return env


def stft_inverse(X: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    """STFT.inverse (stft.py:99-126).

    X: float32 [B, 4, F, T] -> float32 [B, 2, hop*(T-1)] (a [B, 2, F, T] input
    yields the reference's quirky [B, 2, hop*(T-1)/2] fold, tests/unit/test_stft.py:134).
    Bins F..n_fft/2 are zero-padded (stft.py:58-68), planes become complex
    (stft.py:80-97), torch.istft(center=True): irfft (1/n), x window,
    overlap-add, / sum w^2, strip n_fft/2 on both sides.
    """
    X = np.asarray(X, dtype=np.float32)
    B, c4, Fq, T = X.shape
    ch = c4 // 2
    nb = n_fft // 2 + 1
    Xr = X.reshape(B * ch, 2, Fq, T)
    Z = np.zeros((B * ch, T, nb), dtype=np.complex64)
    Z[:, :, :Fq] = (Xr[:, 0] + 1j * Xr[:, 1]).transpose(0, 2, 1)
    y = scipy.fft.irfft(Z, n=n_fft, axis=-1).astype(np.float32)       # [B*ch, T, n]
    y *= hann_periodic(n_fft)[None, None, :]
    total = n_fft + hop * (T - 1)
    acc = np.zeros((B * ch, total), dtype=np.float32)
    for t in range(T):
        acc[:, t * hop: t * hop + n_fft] += y[:, t]
    env = istft_envelope(n_fft, hop, T)
    half = n_fft // 2
    out = acc[:, half: half + hop * (T - 1)] / env[None, half: half + hop * (T - 1)]
    # stft.py:120 reshapes to [*batch, 2, -1] whatever the channel count was
    return out.reshape(B, 2, -1).astype(np.float32)


# --------------------------------------------------------------------------
# chunk loop   (architectures/mdx_separator.py)
# --------------------------------------------------------------------------

@dataclass
class MDXParams:
    """Scalars of MDXSeparator (mdx_separator.py:22-106, 205-228)."""
    n_fft: int = 6144
    hop_length: int = 1024
    dim_f: int = 3072
    segment_size: int = 256
    overlap: float = 0.25
    enable_denoise: bool = False
    compensate: float = 1.0

    @property
    def trim(self) -> int:
        return self.n_fft // 2

    @property
    def chunk_size(self) -> int:
        return self.hop_length * (self.segment_size - 1)


def chunk_plan(N: int, p: MDXParams, is_match_mix: bool = False):
    """Index arithmetic of demix (mdx_separator.py:308-348).

    Returns (chunk_size, gen_size, pad, L, step, starts).
    """
    chunk_size = p.chunk_size
    overlap = 0.02 if is_match_mix else p.overlap
    gen_size = chunk_size - 2 * p.trim
    pad = gen_size + p.trim - (N % gen_size)
    L = p.trim + N + pad
    step = int((1 - overlap) * chunk_size)
    starts = list(range(0, L, step))
    return chunk_size, gen_size, pad, L, step, starts, overlap


def run_model(mix_wave: np.ndarray, p: MDXParams, model_run, is_match_mix=False) -> np.ndarray:
    """MDXSeparator.run_model (mdx_separator.py:414-450): [B,2,C] -> [B,2,C]."""
    spek = stft_forward(mix_wave, p.n_fft, p.hop_length, p.dim_f)
    spek[:, :, :3, :] *= 0                                    # :425
    if is_match_mix:
        spec_pred = spek                                      # :429-432
    elif p.enable_denoise:
        spec_pred = (model_run(-spek) * -0.5) + (model_run(spek) * 0.5)   # :435-440
    else:
        spec_pred = model_run(spek)                           # :443
    return stft_inverse(np.asarray(spec_pred, dtype=np.float32), p.n_fft, p.hop_length)


def demix(mix: np.ndarray, p: MDXParams, model_run, is_match_mix: bool = False) -> np.ndarray:
    """MDXSeparator.demix (mdx_separator.py:293-412): [2,N] f32 -> [2,N] f32."""
    mix = np.asarray(mix, dtype=np.float32)
    N = mix.shape[-1]
    chunk_size, gen_size, pad, L, step, starts, overlap = chunk_plan(N, p, is_match_mix)
    mixture = np.concatenate((np.zeros((2, p.trim), np.float32), mix, np.zeros((2, pad), np.float32)), 1)
    result = np.zeros((1, 2, L), dtype=np.float32)
    divider = np.zeros((1, 2, L), dtype=np.float32)
    for start in starts:
        end = min(start + chunk_size, L)
        n_act = end - start
        window = None
        if overlap != 0:
            window = np.hanning(n_act)                         # float64, symmetric (:358)
            window = np.tile(window[None, None, :], (1, 2, 1))
        part = mixture[:, start:end]
        if end != start + chunk_size:
            part = np.concatenate((part, np.zeros((2, start + chunk_size - end), np.float32)), axis=-1)
        tar = run_model(part[None].astype(np.float32), p, model_run, is_match_mix)
        if window is not None:
            tar[..., :n_act] *= window                         # f32 *= f64 (:387)
            divider[..., start:end] += window                  # :388
        else:
            divider[..., start:end] += 1
        result[..., start:end] += tar[..., :n_act]             # :392
    with np.errstate(divide="ignore", invalid="ignore"):
        tar_waves = result / divider                           # :396 (0/0 at the rim, trimmed below)
    tar_waves = tar_waves[:, :, p.trim:-p.trim]                # :400
    tar_waves = np.concatenate(tar_waves, axis=-1)[:, :N]      # :401
    return tar_waves


def normalize(wave: np.ndarray, max_peak: float = 1.0, min_peak=None) -> np.ndarray:
    """spec_utils.normalize (uvr_lib_v5/spec_utils.py:99-115), in place."""
    maxv = np.abs(wave).max()
    if maxv > max_peak:
        wave *= max_peak / maxv
    elif min_peak is not None and maxv < min_peak:
        wave *= min_peak / maxv
    return wave


def separate_stems(mix: np.ndarray, p: MDXParams, model_run,
                   normalization_threshold: float = 0.9, amplification_threshold: float = 0.0):
    """Stem algebra of MDXSeparator.separate (mdx_separator.py:155-182).

    Returns (primary [N,2], secondary [N,2]); ``mix`` is normalised in place
    exactly like the reference does.
    """
    peak = np.abs(mix).max()
    mix = normalize(mix, normalization_threshold, amplification_threshold)
    source = demix(mix, p, model_run) * peak
    primary = source.T
    secondary = (-primary * p.compensate) + mix.T
    return primary, secondary


# --------------------------------------------------------------------------
# ConvTDFNet   (uvr_lib_v5/mdxnet.py:30-120, uvr_lib_v5/modules.py:5-74)
# --------------------------------------------------------------------------

# NetDims, make_convtdf_state (seeded synthetic weights with the reference class's state_dict names) and synth_mix (the seeded
# synthetic song) live in workload/synth.py -- bench.py's timed workload takes them from there, not from the checker package;
# re-exported here because the oracle's own signatures use them.
from workload.synth import NetDims, make_convtdf_state, synth_mix  # noqa: E402,F401


def _bn(x, sd, prefix):
    """norm(c) of mdxnet.py:45-49: BatchNorm2d in eval mode (optimizer 'rmsprop': the state_dict carries running statistics), or
    GroupNorm(2, c) (optimizer 'adamw': affine only, statistics from the input)."""
    if prefix + ".running_mean" in sd:
        return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                            sd[prefix + ".weight"], sd[prefix + ".bias"], training=False, eps=1e-5)
    return F.group_norm(x, 2, sd[prefix + ".weight"], sd[prefix + ".bias"], eps=1e-5)


def _tfc_tdf(x, sd, prefix, d: NetDims):
    """modules.py:45-74 with TFC (modules.py:5-22), eval mode; bn > 0: two linears, bn == 0: one Linear(f, f) (modules.py:55-60),
    bn is None: no TDF branch.  (DenseTFC, modules.py:25-41, is unreachable: ConvTDFNet never passes dense=True and its forward
    raises -- its convs are c -> c but receive the 2c-channel concatenation.)"""
    for j in range(d.l):
        x = F.conv2d(x, sd[f"{prefix}.tfc.H.{j}.0.weight"], sd[f"{prefix}.tfc.H.{j}.0.bias"], padding=d.k // 2)
        x = F.relu(_bn(x, sd, f"{prefix}.tfc.H.{j}.1"))
    if d.bn is None:
        return x
    t = F.linear(x, sd[f"{prefix}.tdf.0.weight"], sd.get(f"{prefix}.tdf.0.bias"))
    t = F.relu(_bn(t, sd, f"{prefix}.tdf.1"))
    if d.bn != 0:
        t = F.linear(t, sd[f"{prefix}.tdf.3.weight"], sd.get(f"{prefix}.tdf.3.bias"))
        t = F.relu(_bn(t, sd, f"{prefix}.tdf.4"))
    return x + t


@torch.no_grad()
def convtdf_forward(x, sd: dict, d: NetDims):
    """ConvTDFNet.forward (mdxnet.py:97-120), either norm (see _bn),
    eval mode.  x: [B, dim_c, dim_f, dim_t] float32 (numpy or torch) -> same shape (numpy)."""
    was_np = isinstance(x, np.ndarray)
    x = torch.as_tensor(np.ascontiguousarray(x) if was_np else x, dtype=torch.float32)
    x = F.conv2d(x, sd["first_conv.0.weight"], sd["first_conv.0.bias"])
    x = F.relu(_bn(x, sd, "first_conv.1"))
    x = x.transpose(-1, -2)
    skips = []
    for i in range(d.n):
        x = _tfc_tdf(x, sd, f"encoding_blocks.{i}", d)
        skips.append(x)
        x = F.conv2d(x, sd[f"ds.{i}.0.weight"], sd[f"ds.{i}.0.bias"], stride=2)
        x = F.relu(_bn(x, sd, f"ds.{i}.1"))
    x = _tfc_tdf(x, sd, "bottleneck_block", d)
    for i in range(d.n):
        x = F.conv_transpose2d(x, sd[f"us.{i}.0.weight"], sd[f"us.{i}.0.bias"], stride=2)
        x = F.relu(_bn(x, sd, f"us.{i}.1"))
        x = x * skips[-i - 1]
        x = _tfc_tdf(x, sd, f"decoding_blocks.{i}", d)
    x = x.transpose(-1, -2)
    x = F.conv2d(x, sd["final_conv.0.weight"], sd["final_conv.0.bias"])
    return x.numpy() if was_np else x


def make_model_run(sd: dict, d: NetDims):
    """A ``model_run(spek) -> spec_pred`` callable like mdx_separator.py:123."""
    def _run(spek):
        return convtdf_forward(np.asarray(spek, dtype=np.float32), sd, d)
    return _run


def net_flops(d: NetDims, batch: int = 1) -> int:
    """Algorithmic FLOPs (2*MAC) of one ConvTDFNet forward, conv + linear only
    (the same quantity torch.utils.flop_counter counts on the reference class)."""
    fl = 0
    T, Fq, g = d.dim_t, d.dim_f, d.g
    fl += 2 * d.dim_c * g * T * Fq

    def block(c, t, f):
        tdf = 0 if d.bn is None else (2 * c * t * f * f if d.bn == 0 else 2 * 2 * c * t * f * (f // d.bn))
        return d.l * 2 * d.k * d.k * c * c * t * f + tdf
    c, t, f = g, T, Fq
    for _ in range(d.n):
        fl += block(c, t, f)
        fl += 2 * 4 * c * (c + g) * (t // 2) * (f // 2)
        c, t, f = c + g, t // 2, f // 2
    fl += block(c, t, f)
    for _ in range(d.n):
        fl += 2 * 4 * c * (c - g) * t * f
        c, t, f = c - g, t * 2, f * 2
        fl += block(c, t, f)
    fl += 2 * g * d.dim_c * T * Fq
    return fl * batch


# --------------------------------------------------------------------------
# the chunk loop split into its two halves (what one GPU rank computes, and the
# fold rank 0 does after the gather) -- used by the CPU multi-process tests
# --------------------------------------------------------------------------

def demix_chunks(mix: np.ndarray, p: MDXParams, model_run, k0: int, k1: int, is_match_mix: bool = False):
    """Windowed outputs of chunks k0..k1-1 (mdx_separator.py:348-387), zero beyond n_act: [k1-k0, 2, C]."""
    mix = np.asarray(mix, dtype=np.float32)
    N = mix.shape[-1]
    chunk_size, gen_size, pad, L, step, starts, overlap = chunk_plan(N, p, is_match_mix)
    mixture = np.concatenate((np.zeros((2, p.trim), np.float32), mix, np.zeros((2, pad), np.float32)), 1)
    out = np.zeros((k1 - k0, 2, chunk_size), dtype=np.float32)
    for i, k in enumerate(range(k0, k1)):
        start = starts[k]
        end = min(start + chunk_size, L)
        n_act = end - start
        part = np.zeros((2, chunk_size), np.float32)
        part[:, :n_act] = mixture[:, start:end]
        tar = run_model(part[None], p, model_run, is_match_mix)[0]
        if overlap != 0:
            tar[:, :n_act] *= np.hanning(n_act)[None, :]
        out[i, :, :n_act] = tar[:, :n_act]
    return out


def fold_chunks(chunks: np.ndarray, N: int, p: MDXParams, is_match_mix: bool = False) -> np.ndarray:
    """result / divider, trim, [:N] (mdx_separator.py:388-401) from all windowed chunks."""
    chunk_size, gen_size, pad, L, step, starts, overlap = chunk_plan(N, p, is_match_mix)
    result = np.zeros((2, L), dtype=np.float32)
    divider = np.zeros((2, L), dtype=np.float32)
    for k, start in enumerate(starts):
        end = min(start + chunk_size, L)
        n_act = end - start
        if overlap != 0:
            divider[:, start:end] += np.hanning(n_act)[None, :]
        else:
            divider[:, start:end] += 1
        result[:, start:end] += chunks[k][:, :n_act]
    with np.errstate(divide="ignore", invalid="ignore"):
        tar = result / divider
    return tar[:, p.trim:-p.trim][:, :N]
