#!/usr/bin/env python3
"""Golden vectors for the MDXC (TFC-TDF v3) path, produced by the REFERENCE classes themselves.
Build container only (needs /root/reference).  `ml_collections.ConfigDict` is absent here and is
replaced by a minimal attribute-dict with the same access semantics the reference relies on.

    python tests/golden/make_golden_mdxc.py
"""
import importlib.machinery
import logging
import os
import sys
import types

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


def main():
    sys.path.insert(0, ROOT)
    for name in ["onnx", "onnxruntime", "onnx2torch", "librosa", "soundfile", "audioread"]:
        _stub(name)
    _stub("pydub", AudioSegment=object)
    _stub("ml_collections", ConfigDict=ConfigDict)
    for name, path in [("audio_separator", REF + "/audio_separator"),
                       ("audio_separator.separator", REF + "/audio_separator/separator")]:
        pkg = types.ModuleType(name)
        pkg.__path__ = [path]
        pkg.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = pkg
    from audio_separator.separator.architectures.mdxc_separator import MDXCSeparator
    from audio_separator.separator.uvr_lib_v5.tfc_tdf_v3 import TFC_TDF_net
    from oracle import mdxc_oracle as M

    def build(cfg, seed):
        net = TFC_TDF_net(ConfigDict(cfg.as_model_data()), device=torch.device("cpu"))
        sd = M.make_v3_state(cfg, seed)
        missing, unexpected = net.load_state_dict(sd, strict=True), None
        net.eval()
        return net, sd

    out = {}
    # two-stem InstanceNorm/GELU model (the MDX23C shape), small
    cfg2 = M.V3Config(n_fft=128, hop_length=16, dim_f=64, dim_t=16, num_subbands=2, num_scales=2, num_blocks_per_scale=2,
                      num_channels_model=8, growth=8, bottleneck_factor=4)
    net2, _ = build(cfg2, 5)
    rng = np.random.default_rng(61)
    w = (0.4 * rng.standard_normal((2, 2, 240))).astype(np.float32)
    with torch.no_grad():
        out["fwd2"] = net2(torch.tensor(w)).numpy()
    # single-target model (target_instrument set -> one stem + residual)
    cfg1 = M.V3Config(n_fft=128, hop_length=16, dim_f=64, dim_t=16, num_subbands=2, num_scales=2, num_blocks_per_scale=1,
                      num_channels_model=8, growth=4, bottleneck_factor=2, target_instrument="Vocals", act="relu")
    net1, _ = build(cfg1, 6)
    with torch.no_grad():
        out["fwd1"] = net1(torch.tensor(w)).numpy()

    # the TFC branch of MDXCSeparator.demix, reference code
    def ref_demix(net, cfg, mix, overlap, seg=None):
        s = MDXCSeparator.__new__(MDXCSeparator)
        s.logger = logging.getLogger("golden")
        s.pitch_shift = 0
        s.is_roformer = False
        s.model_run = net
        s.torch_device = torch.device("cpu")
        s.model_data_cfgdict = ConfigDict(cfg.as_model_data())
        s.overlap = overlap
        s.batch_size = 2
        s.override_model_segment_size = seg is not None
        s.segment_size = seg
        s.sample_rate = 44100
        s.is_primary_stem_main_target = bool(cfg.target_instrument)
        s.primary_stem_name = cfg.target_instrument or cfg.instruments[0]
        s.secondary_stem_name = "Instrumental"
        return s.demix(mix)

    for name, n in [("n3000", 3000), ("n100", 100), ("n241", 241)]:
        mix = (0.4 * np.random.default_rng(70 + n).standard_normal((2, n))).astype(np.float32)
        d = ref_demix(net2, cfg2, mix, 4)
        out[f"demix2_{name}"] = np.stack([d[k] for k in cfg2.instruments]).astype(np.float32)
    mix = (0.4 * np.random.default_rng(3070).standard_normal((2, 3000))).astype(np.float32)
    d = ref_demix(net2, cfg2, mix, 8, seg=12)
    out["demix2_ov8_seg12"] = np.stack([d[k] for k in cfg2.instruments]).astype(np.float32)
    d = ref_demix(net1, cfg1, mix, 2)
    out["demix1_primary"] = d["Vocals"].astype(np.float32)
    out["demix1_secondary"] = d["Instrumental"].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "mdxc_small.npz"), **out)
    print(os.path.getsize(os.path.join(HERE, "mdxc_small.npz")))


if __name__ == "__main__":
    main()
