#!/usr/bin/env python3
"""Golden vectors for Demucs v3 (HDemucs) from the REFERENCE class (build container only).

    python tests/golden/make_golden_hdemucs.py

Same stubs as make_golden_demucs.py (`julius`, `diffq`: absent third-party deps that are not on the inference path).
"""
import importlib.machinery
import os
import sys
import types

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
for pkg, path in (("audio_separator", f"{REF}/audio_separator"), ("audio_separator.separator", f"{REF}/audio_separator/separator"),
                  ("audio_separator.separator.uvr_lib_v5", f"{REF}/audio_separator/separator/uvr_lib_v5")):
    m = _stub(pkg)
    m.__path__ = [path]

import random  # noqa: E402

from audio_separator.separator.uvr_lib_v5.demucs.apply import apply_model  # noqa: E402
from audio_separator.separator.uvr_lib_v5.demucs.hdemucs import HDemucs  # noqa: E402

sys.path.insert(0, ROOT)
from oracle.hdemucs_oracle import HDConfig, make_hd_state  # noqa: E402


def small_cfg():
    # nfft 1024 -> 512 -> 128 -> 32 -> 8 frequency rows, then the last_freq layer and one time-only layer; norm/LSTM/attention
    # from layer 3 (the released hdemucs_mmi has the same structure one level deeper)
    return HDConfig(channels=8, nfft=1024, depth=5, norm_starts=3, dconv_attn=3, dconv_lstm=3, samplerate=8000, segment=2)


def wide_cfg():
    # dconv_comp 1: DConv hidden sizes 64 / 128 on the two inner levels -> LocalState head dims 16 / 32 (the MFMA attention path)
    return HDConfig(channels=8, nfft=1024, depth=5, norm_starts=3, dconv_attn=3, dconv_lstm=3, dconv_comp=1, samplerate=8000, segment=2)


def main():
    cfg = small_cfg()
    model = HDemucs(**cfg.ctor_kwargs())
    sd = make_hd_state(cfg, 21)
    ref = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    mine = {k: tuple(v.shape) for k, v in sd.items()}
    assert ref == mine, (sorted(set(ref) - set(mine))[:8], sorted(set(mine) - set(ref))[:8],
                         [(k, ref[k], mine[k]) for k in ref if k in mine and ref[k] != mine[k]][:8])
    model.load_state_dict(sd)
    model.eval()
    g = torch.Generator().manual_seed(77)
    out = {}
    # d, e, f: shorter than nfft / than the reflect pad (pad1d's zero extension, hdemucs.py:21-34), down to one frame;
    # 12000 samples: 47 spectrogram frames (no BLSTM framing); 56000: 219 frames > 200 -> the overlapped-frame BLSTM path
    for tag, n in (("a", 12000), ("b", 56000), ("c", 5001), ("d", 877), ("e", 100), ("f", 3)):
        x = torch.randn(1 if tag == "b" else 2, 2, n, generator=g) * 0.3
        with torch.no_grad():
            y = model(x)
        out[f"x_{tag}"] = x.numpy()
        out[f"y_{tag}"] = y.numpy().astype(np.float32)
        print(tag, x.shape, y.shape, float(y.abs().mean()))
    wc = wide_cfg()
    wm = HDemucs(**wc.ctor_kwargs())
    wsd = make_hd_state(wc, 22)
    assert {k: tuple(v.shape) for k, v in wm.state_dict().items()} == {k: tuple(v.shape) for k, v in wsd.items()}
    wm.load_state_dict(wsd)
    wm.eval()
    xw = torch.randn(1, 2, 20000, generator=g) * 0.3
    with torch.no_grad():
        out["x_w"], out["y_w"] = xw.numpy(), wm(xw).numpy().astype(np.float32)
    print("w", xw.shape, float(np.abs(out["y_w"]).mean()))
    # apply_model on 2.3 segments: every chunk runs at its own length (no valid_length), the shift draws are recorded
    mix = torch.randn(1, 2, int(2.3 * cfg.segment * cfg.samplerate) + 77, generator=g) * 0.3
    out["mix"] = mix.numpy()
    with torch.no_grad():
        out["split"] = apply_model(model, mix, shifts=0, split=True, overlap=0.25, progress=False).numpy()
    random.seed(4321)
    offs = []
    real_randint = random.randint

    def rec(a, b):
        offs.append(real_randint(a, b))
        return offs[-1]
    random.randint = rec
    with torch.no_grad():
        out["shift"] = apply_model(model, mix, shifts=2, split=True, overlap=0.25, progress=False).numpy()
    random.randint = real_randint
    out["offsets"] = np.array(offs, np.int64)
    print("apply_model", mix.shape, out["offsets"])
    np.savez_compressed(os.path.join(HERE, "hdemucs_small.npz"), **out)
    print("wrote hdemucs_small.npz", os.path.getsize(os.path.join(HERE, "hdemucs_small.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
