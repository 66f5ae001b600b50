"""CPU oracle for the spectral edges: Ensembler.ensemble (audio_separator/separator/ensembler.py:12-160) and
spec_utils.invert_stem / invert_audio (uvr_lib_v5/spec_utils.py:557-580).

TEST INFRASTRUCTURE ONLY.  librosa.stft / istft are the restatements of oracle/vr_oracle.py (unpinned against librosa
itself); the selection / averaging logic is pinned on golden vectors written by the reference's own Ensembler and
invert_stem driven with those restatements (tests/golden/make_golden_ensemble.py -> ensemble_small.npz).
"""
from __future__ import annotations

import numpy as np

from .vr_oracle import lr_istft, lr_stft

ALGORITHMS = ("avg_wave", "median_wave", "min_wave", "max_wave", "avg_fft", "median_fft", "min_fft", "max_fft", "uvr_max_spec",
              "uvr_min_spec", "ensemble_wav")


def _take(arr, idx):
    return np.squeeze(np.take_along_axis(arr, np.expand_dims(idx, 0), 0), axis=0)


def ensemble(waveforms, algorithm="avg_wave", weights=None):
    """Ensembler.ensemble for equal-length stereo inputs [K][2, N]."""
    waveforms = [np.asarray(w) for w in waveforms]
    if len(waveforms) == 1:
        return waveforms[0]
    weights = np.ones(len(waveforms)) if weights is None else np.array(weights)
    if algorithm == "avg_wave":
        out = np.zeros_like(waveforms[0])
        for w, wt in zip(waveforms, weights):
            out += w * wt
        return out / np.sum(weights)
    if algorithm == "ensemble_wav":
        # spec_utils.ensemble_wav (uvr_lib_v5/spec_utils.py:1245-1266) as Ensembler.ensemble calls it (ensembler.py:71-72): the
        # [channels, N] arrays are split along axis 0 into 240 sections, so section c < channels is channel c (the other 238
        # sections are empty), and each channel is taken whole from the input whose mean |x| over that channel is smallest.
        rows = []
        for c in range(waveforms[0].shape[0]):
            means = [np.abs(w[c]).mean() for w in waveforms]
            rows.append(waveforms[int(np.argmin(means))][c])
        return np.stack(rows)
    if algorithm == "median_wave":
        return np.median(waveforms, axis=0)
    if algorithm == "min_wave":
        a = np.array(waveforms)
        return _take(a, np.argmin(np.abs(a), 0))
    if algorithm == "max_wave":
        a = np.array(waveforms)
        return _take(a, np.argmax(np.abs(a), 0))
    if algorithm in ("avg_fft", "median_fft", "min_fft", "max_fft"):
        n = waveforms[0].shape[-1]
        specs = np.array([lr_stft(w, 2048, 1024) for w in waveforms])
        if algorithm == "avg_fft":
            e = np.zeros_like(specs[0])
            for s_, wt in zip(specs, weights):
                e += s_ * wt
            e /= np.sum(weights)
        elif algorithm == "median_fft":
            e = np.median(np.real(specs), axis=0) + 1j * np.median(np.imag(specs), axis=0)
        elif algorithm == "min_fft":
            e = _take(specs, np.argmin(np.abs(specs), 0))
        else:
            e = _take(specs, np.argmax(np.abs(specs), 0))
        return lr_istft(e, 1024, length=n)
    if algorithm in ("uvr_max_spec", "uvr_min_spec"):
        specs = [lr_stft(w, 2048, 1024) for w in waveforms]
        inp = specs[0]
        for i in range(1, len(specs)):
            ln = min(inp.shape[2], specs[i].shape[2])
            inp, si = inp[:, :, :ln], specs[i][:, :, :ln]
            if algorithm == "uvr_min_spec":
                inp = np.where(np.abs(si) <= np.abs(inp), si, inp)
            else:
                inp = np.where(np.abs(si) >= np.abs(inp), si, inp)
        return lr_istft(inp, 1024, n_fft=2048)
    raise ValueError(f"Unknown ensemble algorithm: {algorithm}")


def invert_stem(mixture, stem):
    """spec_utils.invert_stem (:573-580): mixture, stem [2, N] -> [N', 2]."""
    X = lr_stft(np.asarray(mixture), 2048, 1024)
    Y = lr_stft(np.asarray(stem), 2048, 1024)
    ln = min(X.shape[2], Y.shape[2])
    X, Y = X[:, :, :ln], Y[:, :, :ln]
    max_mag = np.where(np.abs(X) >= np.abs(Y), np.abs(X), np.abs(Y))
    v = Y - max_mag * np.exp(1.0j * np.angle(X))
    return -lr_istft(v, 1024, n_fft=2048).T
