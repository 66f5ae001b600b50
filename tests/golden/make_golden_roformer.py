#!/usr/bin/env python3
"""Golden vectors for the BS-Roformer path from the REFERENCE classes (build container only).

Absent third-party deps are satisfied as follows: `beartype` -> no-op decorator / typing
aliases; `rotary_embedding_torch.RotaryEmbedding` -> the restatement in
oracle/roformer_oracle.py (flagged there); `ml_collections.ConfigDict` -> attribute dict.

    python tests/golden/make_golden_roformer.py
"""
import importlib.machinery
import logging
import os
import sys
import types
import typing

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


class ConfigDict(dict):
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = ConfigDict(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def main():
    sys.path.insert(0, ROOT)
    from oracle import roformer_oracle as R
    for name in ["onnx", "onnxruntime", "onnx2torch", "librosa", "soundfile", "audioread"]:
        _stub(name)
    _stub("pydub", AudioSegment=object)
    _stub("ml_collections", ConfigDict=ConfigDict)
    _stub("beartype", beartype=lambda f: f)
    _stub("beartype.typing", Tuple=typing.Tuple, Optional=typing.Optional, List=typing.List, Callable=typing.Callable)
    _stub("rotary_embedding_torch", RotaryEmbedding=R.RotaryEmbedding)
    for name, path in [("audio_separator", REF + "/audio_separator"),
                       ("audio_separator.separator", REF + "/audio_separator/separator")]:
        pkg = types.ModuleType(name)
        pkg.__path__ = [path]
        pkg.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = pkg
    from audio_separator.separator.uvr_lib_v5.roformer.bs_roformer import BSRoformer
    from audio_separator.separator.architectures.mdxc_separator import MDXCSeparator

    out = {}
    cfg = R.RoformerConfig(dim=32, depth=2, heads=2, dim_head=64, freqs_per_bands=(2, 2, 4, 8, 17), stft_n_fft=64,
                           stft_hop_length=16, stft_win_length=64, dim_t=21, sample_rate=100, mlp_expansion_factor=2)
    sd = R.make_roformer_state(cfg, 7)
    net = BSRoformer(**cfg.model_kwargs(), flash_attn=False)
    net.load_state_dict(sd, strict=True)
    net.eval()
    w = (0.4 * np.random.default_rng(81).standard_normal((2, 2, 320))).astype(np.float32)
    with torch.no_grad():
        out["fwd1"] = net(torch.tensor(w)).numpy()
    # flash path (F.scaled_dot_product_attention) must agree with the einsum path
    netf = BSRoformer(**cfg.model_kwargs(), flash_attn=True)
    netf.load_state_dict(sd, strict=True)
    netf.eval()
    with torch.no_grad():
        out["fwd1_flash"] = netf(torch.tensor(w)).numpy()
    # two stems, deeper transformers
    cfg2 = R.RoformerConfig(dim=32, depth=1, heads=2, dim_head=64, freqs_per_bands=(2, 2, 4, 8, 17), stft_n_fft=64,
                            stft_hop_length=16, stft_win_length=64, dim_t=21, sample_rate=100, num_stems=2,
                            time_transformer_depth=2, freq_transformer_depth=2, target_instrument=None)
    sd2 = R.make_roformer_state(cfg2, 8)
    net2 = BSRoformer(**cfg2.model_kwargs(), flash_attn=False)
    net2.load_state_dict(sd2, strict=True)
    net2.eval()
    with torch.no_grad():
        out["fwd2"] = net2(torch.tensor(w)).numpy()

    # stft_win_length < stft_n_fft: torch.stft / istft zero-pad the Hann window to n_fft at both ends (bs_roformer.py:376-377)
    cfg3 = R.RoformerConfig(dim=32, depth=1, heads=2, dim_head=64, freqs_per_bands=(2, 2, 4, 8, 17), stft_n_fft=64,
                            stft_hop_length=16, stft_win_length=48, dim_t=21, sample_rate=100, mlp_expansion_factor=2)
    sd3 = R.make_roformer_state(cfg3, 9)
    net3 = BSRoformer(**cfg3.model_kwargs(), flash_attn=False)
    net3.load_state_dict(sd3, strict=True)
    net3.eval()
    with torch.no_grad():
        out["fwd3_win48"] = net3(torch.tensor(w)).numpy()

    def ref_demix(model, c, mix, overlap):
        s = MDXCSeparator.__new__(MDXCSeparator)
        s.logger = logging.getLogger("golden")
        s.pitch_shift = 0
        s.is_roformer = True
        s.model_run = model
        s.torch_device = torch.device("cpu")
        s.model_data_cfgdict = ConfigDict(c.as_model_data())
        s.overlap = overlap
        s.batch_size = 1
        s.override_model_segment_size = False
        s.segment_size = None
        s.sample_rate = c.sample_rate
        s.is_primary_stem_main_target = bool(c.target_instrument)
        s.primary_stem_name = c.target_instrument or c.instruments[0]
        s.secondary_stem_name = c.instruments[1]
        return s.demix(mix)

    for name, n, ov in [("n1000_ov8", 1000, 8), ("n1000_ov2", 1000, 2), ("n320_ov1", 320, 1), ("n777_ov2", 777, 2.5)]:
        mix = (0.4 * np.random.default_rng(90 + n).standard_normal((2, n))).astype(np.float32)
        d = ref_demix(net, cfg, mix, ov)
        out[f"demix1_{name}_primary"] = np.asarray(d["vocals"], np.float32)
        out[f"demix1_{name}_secondary"] = np.asarray(d["other"], np.float32)
    mix = (0.4 * np.random.default_rng(1090).standard_normal((2, 1000))).astype(np.float32)
    d = ref_demix(net2, cfg2, mix, 2)
    out["demix2"] = np.stack([d[k] for k in cfg2.instruments]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "roformer_small.npz"), **out)
    print(os.path.getsize(os.path.join(HERE, "roformer_small.npz")))

    # STFT options of the constructor (bs_roformer.py:332-333, 384-386): normalized=True, a non-Hann window function, both with a
    # window shorter than n_fft; forward and the chunk loop
    opts = {}
    for key, kw in (("norm", dict(stft_normalized=True)), ("hamming", dict(stft_window_fn="hamming_window")),
                    ("norm_blackman_win48", dict(stft_normalized=True, stft_window_fn="blackman_window", stft_win_length=48))):
        c = R.RoformerConfig(dim=32, depth=1, heads=2, dim_head=64, freqs_per_bands=(2, 2, 4, 8, 17), stft_n_fft=64, stft_hop_length=16,
                             dim_t=21, sample_rate=100, mlp_expansion_factor=2, **{"stft_win_length": 64, **kw})
        sdo = R.make_roformer_state(c, 10)
        neto = BSRoformer(**c.model_kwargs(), flash_attn=False)
        neto.load_state_dict(sdo, strict=True)
        neto.eval()
        with torch.no_grad():
            opts["fwd_" + key] = neto(torch.tensor(w)).numpy()
        mix = (0.4 * np.random.default_rng(3090).standard_normal((2, 777))).astype(np.float32)
        opts["demix_" + key] = np.asarray(ref_demix(neto, c, mix, 2.5)["vocals"], np.float32)
        print(key, "oracle vs reference", float(np.abs(R.roformer_forward(w, sdo, c) - opts["fwd_" + key]).max()))
    np.savez_compressed(os.path.join(HERE, "roformer_stft_options.npz"), **opts)


if __name__ == "__main__":
    main()
