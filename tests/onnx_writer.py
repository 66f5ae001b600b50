"""Hand-written ONNX files for the reader tests (TEST INFRASTRUCTURE): a protobuf WRITER for the subset of onnx.proto3 that
``audio_separator_amd.onnx_reader`` parses, and a graph builder that lays the ConvTDFNet forward (uvr_lib_v5/mdxnet.py:97-120)
out of an oracle state dict in graph idioms torch's exporter does NOT produce -- real UVR .onnx files come from other exporter
lineages (old opsets, training-mode exports, Gemm-based Linear lowering, NHWC round trips):

  gemm            TDF Linear as Reshape -> Gemm(transB) -> Reshape instead of MatMul (+ Add)
  matmul_add      bias as a separate Add node behind the MatMul
  bn_training     BatchNormalization left unfused, in training-export form: five outputs (Y, running stats, saved stats),
                  training_mode = 1 / a `spatial` attribute, momentum
  opset9          Squeeze / Unsqueeze axes as attributes, initialisers repeated in graph.input, weights as float_data
  transposes      a Transpose pair (NCHW -> NHWC -> NCHW) around every TFC block and an Identity behind every Relu
  shape_ops       the Reshape target of the Gemm lowering computed at run time (Shape -> Gather -> Unsqueeze -> Concat)

Nothing here is shipped; field numbers as in the reader's docstring."""
from __future__ import annotations

import struct

import numpy as np


def _varint(x: int) -> bytes:
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _f(fno: int, wt: int, payload: bytes) -> bytes:
    return _varint((fno << 3) | wt) + (_varint(len(payload)) + payload if wt == 2 else payload)


def _str(fno, s):
    return _f(fno, 2, s.encode())


def tensor(name: str, arr, float_data=False) -> bytes:
    arr = np.asarray(arr)
    b = b"".join(_f(1, 0, _varint(int(d))) for d in arr.shape)
    if arr.dtype.kind == "f":
        a = np.ascontiguousarray(arr, "<f4")
        b += _f(2, 0, _varint(1))
        b += _f(4, 2, a.tobytes()) if float_data else _f(9, 2, a.tobytes())
    else:
        b += _f(2, 0, _varint(7)) + _f(9, 2, np.ascontiguousarray(arr, "<i8").tobytes())
    return b + _str(8, name)


def attr(name, v) -> bytes:
    b = _str(1, name)
    if isinstance(v, float):
        return b + _f(2, 5, struct.pack("<f", v)) + _f(20, 0, _varint(1))
    if isinstance(v, int):
        return b + _f(3, 0, _varint(v)) + _f(20, 0, _varint(2))
    if isinstance(v, (list, tuple)):
        return b + b"".join(_f(8, 0, _varint(int(x))) for x in v) + _f(20, 0, _varint(7))
    if isinstance(v, bytes):                      # a serialised TensorProto (Constant)
        return b + _f(5, 2, v) + _f(20, 0, _varint(4))
    raise TypeError(type(v))


def node(op, inputs, outputs, name="", **attrs) -> bytes:
    b = b"".join(_str(1, i) for i in inputs) + b"".join(_str(2, o) for o in outputs)
    if name:
        b += _str(3, name)
    b += _str(4, op)
    return b + b"".join(_f(5, 2, attr(k, v)) for k, v in attrs.items())


def value_info(name, dims) -> bytes:
    shape = b"".join(_f(1, 2, (_f(1, 0, _varint(d)) if isinstance(d, int) else _str(2, d))) for d in dims)
    ttype = _f(1, 0, _varint(1)) + _f(2, 2, shape)
    return _str(1, name) + _f(2, 2, _f(1, 2, ttype))


def model(nodes, inits, inputs, outputs, opset=13) -> bytes:
    g = b"".join(_f(1, 2, n) for n in nodes) + _str(2, "g") + b"".join(_f(5, 2, t) for t in inits)
    g += b"".join(_f(11, 2, v) for v in inputs) + b"".join(_f(12, 2, v) for v in outputs)
    return _f(1, 0, _varint(7)) + _str(2, "tests/onnx_writer.py") + _f(7, 2, g) + _f(8, 2, _str(1, "") + _f(2, 0, _varint(opset)))


def convtdf_graph(sd: dict, d, idioms=(), extra_node=None) -> bytes:
    """ConvTDFNet.forward as an ONNX graph over UNFOLDED parameters (every BatchNorm its own node), written in the given idioms.
    ``extra_node``: (op, after) -- splice an activation-side node of that op behind the first Relu (for the rejection tests)."""
    I = set(idioms)
    nodes, inits, init_names = [], [], []
    cnt = [0]

    def nm(p):
        cnt[0] += 1
        return f"{p}_{cnt[0]}"

    def init(name, arr):
        inits.append(tensor(name, np.asarray(arr), float_data=("opset9" in I)))
        init_names.append((name, list(np.asarray(arr).shape)))
        return name

    def t(key):
        v = sd[key]
        return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)

    def bn(x, prefix):
        ins = [x] + [init(f"{prefix}.{k}", t(f"{prefix}.{k}")) for k in ("weight", "bias", "running_mean", "running_var")]
        y = nm("bn")
        if "bn_training" in I:
            outs = [y] + [nm("bn_aux") for _ in range(4)]
            kw = dict(epsilon=1e-5, momentum=0.9)
            kw.update({"spatial": 1} if "opset9" in I else {"training_mode": 1})
            nodes.append(node("BatchNormalization", ins, outs, name=nm("BatchNormalization"), **kw))
        else:
            nodes.append(node("BatchNormalization", ins, [y], name=nm("BatchNormalization"), epsilon=1e-5))
        return y

    def relu(x):
        y = nm("relu")
        nodes.append(node("Relu", [x], [y], name=nm("Relu")))
        if "transposes" in I:
            z = nm("id")
            nodes.append(node("Identity", [y], [z]))
            y = z
        return y

    def conv(x, prefix, k, stride=1, pad=0, transposed=False):
        w, b = init(f"{prefix}.weight", t(f"{prefix}.weight")), init(f"{prefix}.bias", t(f"{prefix}.bias"))
        y = nm("conv")
        nodes.append(node("ConvTranspose" if transposed else "Conv", [x, w, b], [y], name=nm("Conv"), kernel_shape=[k, k],
                          strides=[stride, stride], pads=[pad] * 4, dilations=[1, 1], group=1))
        return y

    def linear(x, prefix, c, tt, fin, fout):
        w = t(f"{prefix}.weight")                       # [fout, fin]
        has_b = f"{prefix}.bias" in sd
        if "gemm" in I:
            if "shape_ops" in I:                        # [-1, fin] / [B, c, T, fout] assembled from Shape(x) at run time
                shp, g0, u0 = nm("shape"), nm("gather"), nm("unsq")
                nodes.append(node("Shape", [x], [shp]))
                nodes.append(node("Gather", [shp, init(nm("idx"), np.asarray(0, np.int64))], [g0], axis=0))
                if "opset9" in I:
                    nodes.append(node("Unsqueeze", [g0], [u0], axes=[0]))
                else:
                    nodes.append(node("Unsqueeze", [g0, init(nm("axes"), np.asarray([0], np.int64))], [u0]))
                back = nm("concat")
                nodes.append(node("Concat", [u0, init(nm("tail"), np.asarray([c, tt, fout], np.int64))], [back], axis=0))
            else:
                back = init(nm("shape_back"), np.asarray([0, c, tt, fout], np.int64))
            flat, y2 = nm("flat"), nm("gemm")
            nodes.append(node("Reshape", [x, init(nm("shape2d"), np.asarray([-1, fin], np.int64))], [flat]))
            ins = [flat, init(f"{prefix}.weight", w)] + ([init(f"{prefix}.bias", t(f"{prefix}.bias"))] if has_b else [])
            nodes.append(node("Gemm", ins, [y2], name=nm("Gemm"), alpha=1.0, beta=1.0, transB=1))
            y = nm("unflat")
            nodes.append(node("Reshape", [y2, back], [y]))
            return y
        y = nm("mm")
        nodes.append(node("MatMul", [x, init(f"{prefix}.weight_t", np.ascontiguousarray(w.T))], [y], name=nm("MatMul")))
        if has_b:
            z = nm("add")
            nodes.append(node("Add", [y, init(f"{prefix}.bias", t(f"{prefix}.bias"))], [z], name=nm("Add")))
            y = z
        return y

    def transpose(x, perm):
        y = nm("tr")
        nodes.append(node("Transpose", [x], [y], perm=list(perm)))
        return y

    def tfc_tdf(x, prefix, c, tt, f):
        if "transposes" in I:                           # an NHWC round trip in front of the block
            x = transpose(transpose(x, (0, 2, 3, 1)), (0, 3, 1, 2))
        for j in range(d.l):
            x = relu(bn(conv(x, f"{prefix}.tfc.H.{j}.0", 3, pad=1), f"{prefix}.tfc.H.{j}.1"))
        h = relu(bn(linear(x, f"{prefix}.tdf.0", c, tt, f, f // d.bn), f"{prefix}.tdf.1"))
        h = relu(bn(linear(h, f"{prefix}.tdf.3", c, tt, f // d.bn, f), f"{prefix}.tdf.4"))
        y = nm("res")
        nodes.append(node("Add", [x, h], [y]))
        return y

    g, n = d.g, d.n
    x = relu(bn(conv("input", "first_conv.0", 1), "first_conv.1"))
    if extra_node is not None:
        y = nm("extra")
        nodes.append(node(extra_node, [x], [y], name="unexpected_" + extra_node))
        x = y
    x = transpose(x, (0, 1, 3, 2))
    c, tt, f = g, d.dim_t, d.dim_f
    skips = []
    for i in range(n):
        x = tfc_tdf(x, f"encoding_blocks.{i}", c, tt, f)
        skips.append(x)
        x = relu(bn(conv(x, f"ds.{i}.0", 2, stride=2), f"ds.{i}.1"))
        c, tt, f = c + g, tt // 2, f // 2
    x = tfc_tdf(x, "bottleneck_block", c, tt, f)
    for i in range(n):
        x = relu(bn(conv(x, f"us.{i}.0", 2, stride=2, transposed=True), f"us.{i}.1"))
        c, tt, f = c - g, tt * 2, f * 2
        y = nm("mul")
        nodes.append(node("Mul", [x, skips[-i - 1]], [y]))
        x = tfc_tdf(y, f"decoding_blocks.{i}", c, tt, f)
    x = transpose(x, (0, 1, 3, 2))
    w, b = init("final_conv.0.weight", t("final_conv.0.weight")), init("final_conv.0.bias", t("final_conv.0.bias"))
    nodes.append(node("Conv", [x, w, b], ["output"], name="final", kernel_shape=[1, 1], strides=[1, 1], pads=[0, 0, 0, 0]))
    inputs = [value_info("input", [1, d.dim_c, d.dim_f, d.dim_t])]
    if "opset9" in I:                                   # IR version < 4: every initialiser is ALSO a graph input
        inputs += [value_info(name, dims) for name, dims in init_names]
    return model(nodes, inits, inputs, [value_info("output", [1, d.dim_c, d.dim_f, d.dim_t])], opset=9 if "opset9" in I else 13)
