#!/usr/bin/env python3
"""Golden vectors for the Demucs v4 path from the REFERENCE classes (build container only).

Absent third-party deps: `julius` (only used by the v1-v3 Demucs resampler) and `diffq` (only
used for quantised checkpoints) are satisfied with empty stub modules; neither is on the
HTDemucs inference path.

    python tests/golden/make_golden_demucs.py
"""
import importlib.machinery
import os
import random
import sys
import types
from fractions import Fraction

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_stub("julius")
_stub("diffq", UniformQuantizer=object, DiffQuantizer=object, restore_quantized_state=lambda *a, **k: None)
# import the demucs sub-package without executing audio_separator/__init__ (it pulls in the CLI deps)
for pkg, path in (("audio_separator", f"{REF}/audio_separator"), ("audio_separator.separator", f"{REF}/audio_separator/separator"),
                  ("audio_separator.separator.uvr_lib_v5", f"{REF}/audio_separator/separator/uvr_lib_v5")):
    m = _stub(pkg)
    m.__path__ = [path]

from audio_separator.separator.uvr_lib_v5.demucs.htdemucs import HTDemucs  # noqa: E402
from audio_separator.separator.uvr_lib_v5.demucs.apply import apply_model  # noqa: E402

sys.path.insert(0, ROOT)
from oracle.demucs_oracle import HTConfig, make_ht_state  # noqa: E402


def small_cfg():
    # 4 sources, nfft 1024 (hop 256), depth 3 -> 8 freq rows at the bottom; transformer dim 128 via
    # bottom_channels, 2 heads -> head dim 64; segment 1.0 s at 8 kHz
    return HTConfig(channels=16, nfft=1024, depth=3, bottom_channels=128, t_layers=3, t_heads=2,
                    samplerate=8000, segment=Fraction(1, 1))


def small_cfg48():
    # head dim 48 like the released htdemucs (384 / 8); t_layers 2
    return HTConfig(channels=24, nfft=1024, depth=3, bottom_channels=0, t_layers=2, t_heads=2,
                    samplerate=8000, segment=Fraction(1, 1))


def build(cfg, seed):
    model = HTDemucs(**cfg.ctor_kwargs())
    sd = make_ht_state(cfg, seed)
    ref_keys = set(model.state_dict().keys())
    assert ref_keys == set(sd.keys()), (sorted(ref_keys - set(sd))[:8], sorted(set(sd) - ref_keys)[:8])
    model.load_state_dict(sd)
    return model.eval(), sd


def main():
    torch.manual_seed(0)
    out = {}
    for tag, cfg, seed in (("a", small_cfg(), 11), ("b", small_cfg48(), 12)):
        model, sd = build(cfg, seed)
        g = torch.Generator().manual_seed(100 + seed)
        tl = cfg.training_length
        x = torch.randn(2, 2, tl, generator=g) * 0.3
        with torch.no_grad():
            y = model(x)
        out[f"{tag}_fwd_in"] = x.numpy()
        out[f"{tag}_fwd_out"] = y.numpy()
        # shorter than the training segment: the model pads internally (htdemucs.py:497-502)
        xs = x[:1, :, : tl - 777]
        with torch.no_grad():
            out[f"{tag}_short_out"] = model(xs).numpy()
        if tag == "a":
            # apply_model: split + shifts with the random draws captured
            L = int(2.6 * tl) + 123
            mix = torch.randn(1, 2, L, generator=g) * 0.3
            out["a_mix"] = mix.numpy()
            with torch.no_grad():
                out["a_split"] = apply_model(model, mix, shifts=0, split=True, overlap=0.25, progress=False).numpy()
            # record the draws: the model itself also consumes `random` (transformer.py:509 randrange), so the
            # offsets cannot be predicted from the seed alone
            random.seed(1234)
            offs = []
            real_randint = random.randint

            def rec(a, b):
                offs.append(real_randint(a, b))
                return offs[-1]
            random.randint = rec
            with torch.no_grad():
                out["a_shift"] = apply_model(model, mix, shifts=2, split=True, overlap=0.25, progress=False).numpy()
            random.randint = real_randint
            out["a_offsets"] = np.array(offs, np.int64)
            with torch.no_grad():
                out["a_nosplit"] = apply_model(model, mix[..., : tl - 100], shifts=0, split=False, progress=False).numpy()
    np.savez_compressed(os.path.join(HERE, "demucs_small.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
