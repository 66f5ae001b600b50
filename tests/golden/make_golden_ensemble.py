#!/usr/bin/env python3
"""Golden vectors for the spectral edges from the REFERENCE Ensembler and spec_utils.invert_stem (build container only).
`librosa` -> stand-in module with the stft / istft restatements of oracle/vr_oracle.py.

    python tests/golden/make_golden_ensemble.py
"""
import importlib.machinery
import logging
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import vr_oracle as V  # noqa: E402
from oracle.ensemble_oracle import ALGORITHMS  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_stub("librosa", stft=V.lr_stft, istft=V.lr_istft, resample=V.lr_resample)
_stub("soundfile")
_stub("audioread")
for pkg, path in (("audio_separator", f"{REF}/audio_separator"), ("audio_separator.separator", f"{REF}/audio_separator/separator"),
                  ("audio_separator.separator.uvr_lib_v5", f"{REF}/audio_separator/separator/uvr_lib_v5")):
    m = _stub(pkg)
    m.__path__ = [path]

from audio_separator.separator.ensembler import Ensembler  # noqa: E402
from audio_separator.separator.uvr_lib_v5 import spec_utils  # noqa: E402


def main():
    rng = np.random.default_rng(21)
    n = 9001
    t = np.arange(n) / 8000.0
    waves = [np.stack([0.3 * np.sin(2 * np.pi * (200 + 40 * k) * t + k), 0.2 * np.cos(2 * np.pi * (310 + 25 * k) * t)]) +
             0.1 * rng.standard_normal((2, n)) for k in range(4)]
    waves = [w.astype(np.float32) for w in waves]
    out = {"waves": np.stack(waves)}
    log = logging.getLogger("golden")
    for alg in ALGORITHMS:
        out[f"{alg}_k4"] = np.asarray(Ensembler(log, alg, None).ensemble([w.copy() for w in waves]))
        out[f"{alg}_k3"] = np.asarray(Ensembler(log, alg, None).ensemble([w.copy() for w in waves[:3]]))
    for alg in ("avg_wave", "avg_fft"):
        out[f"{alg}_w"] = np.asarray(Ensembler(log, alg, [1.0, 2.0, 0.5, 0.25]).ensemble([w.copy() for w in waves]))
    out["invert"] = spec_utils.invert_stem(waves[0].copy(), waves[1].copy())
    np.savez_compressed(os.path.join(HERE, "ensemble_small.npz"), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
