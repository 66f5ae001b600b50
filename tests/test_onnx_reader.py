"""ONNX reader (model-file loading stays drop-in) against .onnx files written by
torch's own exporter from the reference ConvTDFNet class (tests/golden/make_onnx_fixture.py)."""
import os

import numpy as np
import pytest

from oracle import mdx_oracle as O
from audio_separator_amd.onnx_reader import OnnxFormatError, convtdf_from_onnx, parse_onnx
from audio_separator_amd.weights import fold_convtdf_state


@pytest.mark.parametrize("fname,bias,seed", [("net_small.onnx", False, 3), ("net_small_bias.onnx", True, 4)])
def test_reader_recovers_config_and_weights(golden_dir, fname, bias, seed):
    cfg, tensors = convtdf_from_onnx(os.path.join(golden_dir, fname))
    assert (cfg.dim_c, cfg.dim_f, cfg.dim_t, cfg.g, cfg.l, cfg.num_blocks, cfg.k, cfg.bn, cfg.tdf_bias) == \
        (4, 32, 16, 8, 2, 5, 3, 4, bias)
    d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4, bias=bias)
    want = fold_convtdf_state(O.make_convtdf_state(d, seed=seed), d.num_blocks, d.l, tdf_bias=bias)
    assert sorted(tensors) == sorted(want)
    for k in want:
        assert tensors[k].shape == want[k].shape, k
        assert tensors[k].dtype == np.float32
        # the exporter folds conv+BN in float32, we fold in float64
        assert np.allclose(tensors[k], want[k], rtol=2e-6, atol=2e-7), k


@pytest.mark.parametrize("fname", ["net_small_bias_op11.onnx", "net_small_bias_op17_nofold.onnx"])
def test_reader_other_exporter_settings(golden_dir, fname):
    """opset 11 with fused Conv+BN, and opset 17 exported WITHOUT constant folding: 25 BatchNormalization nodes (one after every
    Conv / ConvTranspose / MatMul), every Linear weight behind a Transpose node, dynamic batch / time axes."""
    path = os.path.join(golden_dir, fname)
    nodes, inits, inputs = parse_onnx(path)
    ops = [n.op for n in nodes]
    if "nofold" in fname:
        assert ops.count("BatchNormalization") == 25 and ops.count("Transpose") == 12 and inputs["input"] == [None, 4, 32, None]
        with pytest.raises(OnnxFormatError, match="time size"):
            convtdf_from_onnx(path)                      # dynamic axis: the caller supplies dim_t (= segment_size)
    cfg, tensors = convtdf_from_onnx(path, dim_t=16)
    assert (cfg.dim_c, cfg.dim_f, cfg.dim_t, cfg.g, cfg.l, cfg.num_blocks, cfg.k, cfg.bn, cfg.tdf_bias) == (4, 32, 16, 8, 2, 5, 3, 4, True)
    d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4, bias=True)
    want = fold_convtdf_state(O.make_convtdf_state(d, seed=4), d.num_blocks, d.l, tdf_bias=True)
    assert sorted(tensors) == sorted(want)
    for k in want:
        assert np.allclose(tensors[k], want[k], rtol=2e-6, atol=2e-7), k


@pytest.mark.parametrize("tag,norm,bn,bias,seed", [("gn_bias", "group", 4, True, 12), ("gn_op17_nofold", "group", 4, False, 11), ("gn_bn0", "group", 0, True, 14),
                                                   ("bn0", "batch", 0, False, 13), ("notdf", "batch", None, False, 15)])
def test_reader_convtdf_variants(golden_dir, tag, norm, bn, bias, seed):
    """The other ConvTDFNet forms the reference class builds (uvr_lib_v5/mdxnet.py:45-49, modules.py:52-70), exported by torch from
    that class: GroupNorm(2, c) arrives as Reshape -> InstanceNormalization -> Reshape -> Mul -> Add and is read back as UNFOLDED
    weights + the affine of every norm; bn == 0 has one square TDF linear per block; bn is None has none."""
    path = os.path.join(golden_dir, f"net_small_{tag}.onnx")
    cfg, tensors = convtdf_from_onnx(path, dim_t=16)
    assert (cfg.dim_c, cfg.dim_f, cfg.dim_t, cfg.g, cfg.l, cfg.num_blocks, cfg.k, cfg.bn, cfg.tdf_bias, cfg.norm) == (4, 32, 16, 8, 2, 5, 3, bn, bias, norm)
    d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=bn, bias=bias, norm=norm)
    want = fold_convtdf_state(O.make_convtdf_state(d, seed=seed), d.num_blocks, d.l, tdf_bias=bias)
    assert sorted(tensors) == sorted(want)
    if norm == "group":
        # one norm behind every conv / linear but the final conv: 5 blocks x (2 TFC + 2 or 1 TDF) + first + 2 ds + 2 us
        assert sum(k.endswith(".gn_w") for k in tensors) == (25 if bn else 20) and not any(k.endswith(".scale") for k in tensors)
    for k in want:
        assert tensors[k].shape == want[k].shape and tensors[k].dtype == np.float32, k
        assert np.allclose(tensors[k], want[k], rtol=2e-6, atol=2e-7), k


def test_graph_structure(golden_dir):
    nodes, inits, inputs = parse_onnx(os.path.join(golden_dir, "net_small.onnx"))
    ops = [n.op for n in nodes]
    assert ops[0] == "Conv" and ops[-1] == "Conv" and ops.count("ConvTranspose") == 2 and ops.count("MatMul") == 10
    assert inputs["input"] == [1, 4, 32, 16]
    assert all(a.dtype in (np.float32, np.int64) for a in inits.values())


def test_dim_t_override(golden_dir):
    cfg, _ = convtdf_from_onnx(os.path.join(golden_dir, "net_small.onnx"), dim_t=32)
    assert cfg.dim_t == 32


def test_rejects_garbage(tmp_path):
    p = tmp_path / "x.onnx"
    p.write_bytes(b"\x08\x01\x12\x03abc")          # a valid protobuf without a graph
    with pytest.raises(OnnxFormatError):
        convtdf_from_onnx(str(p))


# ---- graph idioms torch's exporter does not produce (hand-written files: tests/onnx_writer.py) --------------------------------
from tests import onnx_writer as W  # noqa: E402

IDIOMS = [(), ("gemm",), ("gemm", "shape_ops"), ("bn_training",), ("opset9",), ("opset9", "bn_training", "gemm", "shape_ops"),
          ("transposes",), ("gemm", "transposes", "bn_training", "opset9")]


@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("idioms", IDIOMS, ids=["+".join(i) or "plain" for i in IDIOMS])
def test_reader_other_exporter_lineages(idioms, bias):
    """Gemm-lowered Linear (Reshape -> Gemm(transB) -> Reshape, with the Reshape target from run-time Shape / Gather / Concat),
    unfused BatchNormalization in training-export form (five outputs, training_mode / spatial / momentum attributes), opset 9
    (initialisers repeated in graph.input, float_data payloads, axes attributes), NHWC Transpose pairs and Identity nodes on the
    activations: the reader must recover exactly what folding the state dict in float64 gives."""
    d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4, bias=bias)
    sd = O.make_convtdf_state(d, seed=4)
    blob = W.convtdf_graph(sd, d, idioms)
    nodes, inits, inputs = parse_onnx(blob)
    ops = [n.op for n in nodes]
    if "gemm" in idioms:
        assert ops.count("Gemm") == 10 and "MatMul" not in ops
    if "bn_training" in idioms:
        assert all(len(n.outputs) == 5 for n in nodes if n.op == "BatchNormalization")
    if "opset9" in idioms:
        assert len(inputs) > 100 and inputs["input"] == [1, 4, 32, 16]        # every initialiser is also listed as a graph input
    if "transposes" in idioms:
        assert ops.count("Transpose") == 2 + 2 * 5 and ops.count("Identity") == ops.count("Relu")
    cfg, tensors = convtdf_from_onnx(blob)
    assert (cfg.dim_c, cfg.dim_f, cfg.dim_t, cfg.g, cfg.l, cfg.num_blocks, cfg.k, cfg.bn, cfg.tdf_bias) == (4, 32, 16, 8, 2, 5, 3, 4, bias)
    want = fold_convtdf_state(sd, d.num_blocks, d.l, tdf_bias=bias)
    assert sorted(tensors) == sorted(want)
    for k in want:
        assert tensors[k].shape == want[k].shape and np.array_equal(tensors[k], want[k]), k


@pytest.mark.parametrize("op,what", [("Sigmoid", "unsupported op"), ("LeakyRelu", "unsupported op"), ("Pad", "unsupported op"),
                                     ("InstanceNormalization", "outside the GroupNorm"), ("Slice", "shape arithmetic"), ("Squeeze", "activation")])
def test_reader_fails_loudly_on_ops_it_cannot_account_for(op, what):
    """An activation-side node that is not part of the BatchNorm ConvTDFNet must stop the load, with the node named -- skipping
    it would silently run a different model."""
    d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4)
    sd = O.make_convtdf_state(d, seed=4)
    with pytest.raises(OnnxFormatError, match=f"unexpected_{op}.*{what}"):
        convtdf_from_onnx(W.convtdf_graph(sd, d, (), extra_node=op))


def test_writer_files_equal_exporter_files_semantically(golden_dir):
    """The hand-written plain graph and torch's export of the same net give the same engine tensors (exporter: float32 fold)."""
    d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4, bias=True)
    sd = O.make_convtdf_state(d, seed=4)
    _, mine = convtdf_from_onnx(W.convtdf_graph(sd, d, ()))
    _, theirs = convtdf_from_onnx(os.path.join(golden_dir, "net_small_bias.onnx"))
    assert sorted(mine) == sorted(theirs)
    for k in mine:
        assert np.allclose(mine[k], theirs[k], rtol=2e-6, atol=2e-7), k
