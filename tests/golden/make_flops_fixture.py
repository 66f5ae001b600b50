#!/usr/bin/env python3
"""FLOP counts of the REFERENCE net classes (torch.utils.flop_counter.FlopCounterMode, conv + matmul, 2 x MAC), stored as
tests/golden/flops.json and compared with the engine's own closed-form counters (asx_net_flops / asx_v3_flops), which feed every
roofline figure bench.py prints.  Build container only (needs /root/reference).

    python tests/golden/make_flops_fixture.py
"""
import json
import os
import sys

import torch
from torch.utils.flop_counter import FlopCounterMode

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden_mdxc as G  # noqa: E402  (stubs + package shims for the reference imports)


def main():
    for name in ["onnx", "onnxruntime", "onnx2torch", "librosa", "soundfile", "audioread"]:
        G._stub(name)
    G._stub("pydub", AudioSegment=object)
    G._stub("ml_collections", ConfigDict=G.ConfigDict)
    G._stub("pytorch_lightning", LightningModule=torch.nn.Module)
    import importlib.machinery
    import types
    for name, path in [("audio_separator", G.REF + "/audio_separator"), ("audio_separator.separator", G.REF + "/audio_separator/separator")]:
        pkg = types.ModuleType(name)
        pkg.__path__ = [path]
        pkg.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = pkg
    from audio_separator.separator.uvr_lib_v5.tfc_tdf_v3 import TFC_TDF_net
    from oracle import mdxc_oracle as M
    out = {"what": "FlopCounterMode totals (conv + mm/bmm/addmm, 2 x MAC) of the reference classes, one chunk, batch 1"}

    def v3(cfg):
        net = TFC_TDF_net(G.ConfigDict(cfg.as_model_data()), device=torch.device("cpu")).eval()
        chunk = cfg.hop_length * (cfg.dim_t - 1)
        x = torch.zeros(1, 2, chunk)
        with torch.no_grad(), FlopCounterMode(display=False) as fc:
            net(x)
        return int(fc.get_total_flops())

    # tests/test_gpu_mdxc.py::test_mdx23c_shape_excerpt_vs_oracle
    cfg = M.V3Config(n_fft=8192, hop_length=1024, dim_f=4096, dim_t=64, num_subbands=4, num_scales=3,
                     num_blocks_per_scale=2, num_channels_model=32, growth=32, bottleneck_factor=4)
    out["mdx23c_excerpt"] = v3(cfg)
    cfg2 = M.V3Config(n_fft=128, hop_length=16, dim_f=64, dim_t=16, num_subbands=2, num_scales=2, num_blocks_per_scale=2,
                      num_channels_model=8, growth=8, bottleneck_factor=4)
    out["mdxc_small_cfg2"] = v3(cfg2)

    # ConvTDFNet (mdxnet.py:30) on the HQ_3 geometry and on the small golden geometry
    from audio_separator.separator.uvr_lib_v5.mdxnet import ConvTDFNet

    def convtdf(dim_f, dim_t_log2, g, l, nb, bn, n_fft, hop):
        net = ConvTDFNet(target_name="vocals", lr=1e-3, optimizer="rmsprop", dim_c=4, dim_f=dim_f, dim_t=2 ** dim_t_log2, n_fft=n_fft,
                         hop_length=hop, num_blocks=nb, l=l, g=g, k=3, bn=bn, bias=False, overlap=0).eval()
        x = torch.zeros(1, 4, dim_f, 2 ** dim_t_log2)
        with torch.no_grad(), FlopCounterMode(display=False) as fc:
            net(x)
        return int(fc.get_total_flops())

    out["convtdf_hq3_g48"] = convtdf(3072, 8, 48, 3, 11, 8, 6144, 1024)
    out["convtdf_hq3_g8"] = convtdf(3072, 8, 8, 3, 11, 8, 6144, 1024)
    with open(os.path.join(HERE, "flops.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(out)


if __name__ == "__main__":
    main()
