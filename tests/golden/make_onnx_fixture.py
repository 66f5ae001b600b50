#!/usr/bin/env python3
"""Export the REFERENCE ConvTDFNet (uvr_lib_v5/mdxnet.py) to real .onnx files with torch's
TorchScript exporter, for the ONNX reader tests.  Build container only (needs /root/reference).

torch.onnx.export(dynamo=False) serialises the ModelProto in C++ and only then calls into the
`onnx` package for an onnxscript post-pass; that package is absent here, so the post-pass is
replaced by the identity.  The bytes written are what the exporter produced.

    python tests/golden/make_onnx_fixture.py
"""
import importlib.machinery
import os
import sys
import types
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
warnings.filterwarnings("ignore")


def main():
    sys.path.insert(0, ROOT)
    m = types.ModuleType("pytorch_lightning")
    m.__spec__ = importlib.machinery.ModuleSpec("pytorch_lightning", None)
    m.LightningModule = torch.nn.Module
    sys.modules["pytorch_lightning"] = m
    for name, path in [("audio_separator", REF + "/audio_separator"),
                       ("audio_separator.separator", REF + "/audio_separator/separator"),
                       ("audio_separator.separator.uvr_lib_v5", REF + "/audio_separator/separator/uvr_lib_v5")]:
        pkg = types.ModuleType(name)
        pkg.__path__ = [path]
        pkg.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = pkg
    from audio_separator.separator.uvr_lib_v5.mdxnet import ConvTDFNet
    from oracle import mdx_oracle as O
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto

    for fname, bias, seed in [("net_small.onnx", False, 3), ("net_small_bias.onnx", True, 4)]:
        d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4, bias=bias)
        sd = O.make_convtdf_state(d, seed=seed)
        net = ConvTDFNet("t", 1e-3, "rmsprop", 4, 32, 16, 96, 16, 5, 2, 8, 3, 4, bias, 0)
        net.load_state_dict(sd, strict=False)
        net.eval()
        out = os.path.join(HERE, fname)
        torch.onnx.export(net, (torch.randn(1, 4, 32, 16),), out, input_names=["input"], output_names=["output"],
                          dynamo=False, opset_version=13)
        print(out, os.path.getsize(out))
        if bias:
            # the same net through other exporter settings real MDX-Net files were written with: an older opset, and no
            # constant folding (BatchNormalization left unfused after every Conv, Linear weights behind Transpose nodes,
            # dynamic batch / time axes so that the time size is NOT recorded in the graph input)
            for tag, kw in (("op11", dict(opset_version=11)),
                            ("op17_nofold", dict(opset_version=17, do_constant_folding=False,
                                                 dynamic_axes={"input": {0: "b", 3: "t"}, "output": {0: "b", 3: "t"}}))):
                out2 = os.path.join(HERE, fname.replace(".onnx", f"_{tag}.onnx"))
                torch.onnx.export(net, (torch.randn(1, 4, 32, 16),), out2, input_names=["input"], output_names=["output"],
                                  dynamo=False, **kw)
                print(out2, os.path.getsize(out2))

    # the ConvTDFNet forms besides BatchNorm / bn > 0 (tests/golden/make_golden_variants.py holds their outputs): GroupNorm(2, c) --
    # lowered by the exporter to Reshape -> InstanceNormalization -> Reshape -> Mul -> Add --, bn == 0, bn is None
    variants = {"gn_bias": ("adamw", 4, True, 12, dict(opset_version=13)),
                "gn_op17_nofold": ("adamw", 4, False, 11, dict(opset_version=17, do_constant_folding=False,
                                                               dynamic_axes={"input": {0: "b", 3: "t"}, "output": {0: "b", 3: "t"}})),
                "gn_bn0": ("adamw", 0, True, 14, dict(opset_version=13)), "bn0": ("rmsprop", 0, False, 13, dict(opset_version=13)),
                "notdf": ("rmsprop", None, False, 15, dict(opset_version=13))}
    for tag, (opt, bn, bias, seed, kw) in variants.items():
        d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=bn, bias=bias, norm="group" if opt == "adamw" else "batch")
        net = ConvTDFNet("t", 1e-3, opt, 4, 32, 16, 96, 16, 5, 2, 8, 3, bn, bias, 0)
        net.load_state_dict(O.make_convtdf_state(d, seed=seed), strict=False)
        net.eval()
        out = os.path.join(HERE, f"net_small_{tag}.onnx")
        torch.onnx.export(net, (torch.randn(1, 4, 32, 16),), out, input_names=["input"], output_names=["output"], dynamo=False, **kw)
        print(out, os.path.getsize(out))


if __name__ == "__main__":
    main()
