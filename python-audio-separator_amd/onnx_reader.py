"""Minimal ONNX reader for MDX-Net (ConvTDFNet) model files.

The reference hands ``model_path`` to ``ort.InferenceSession`` (mdx_separator.py:122)
or ``onnx2torch.convert`` (:126-132).  This module replaces both for the weight
side: it parses the protobuf wire format of an ``.onnx`` file directly (the image
has neither ``onnx`` nor ``onnxruntime``), walks ``graph.node`` in execution order
and maps the ConvTDFNet pattern (uvr_lib_v5/mdxnet.py:54-120) onto the engine's
canonical tensors (include/asx.h), folding BatchNormalization nodes in float64.

Only the ONNX features such exports use are understood: Conv, ConvTranspose,
MatMul (+ optional bias Add) or Gemm (behind a Reshape pair) for the TDF linears,
BatchNormalization (fused into the Conv by the exporter or left as its own node, in
inference or training-export form), Relu, Add, Mul, Transpose / Identity / Dropout on
activations, and the shape arithmetic of run-time Reshape targets (Shape, Gather,
Concat, Slice ... on shape vectors only); weights as fp32 initialisers (``raw_data``
or ``float_data``, also when repeated in ``graph.input`` as old IR versions do) or
derived from them through Constant / Identity / Transpose / Reshape / Squeeze /
Unsqueeze / Cast nodes (exports made without constant folding).  EVERY node of the
graph must be one of those: anything else raises ``OnnxFormatError`` naming the node
rather than being skipped (``check_ops``) -- an activation-side op the engine does not
run would otherwise silently change the model.

Field numbers follow onnx.proto3 (ModelProto.graph = 7; GraphProto.node = 1,
initializer = 5, input = 11; NodeProto.input = 1, output = 2, name = 3, op_type = 4,
attribute = 5; AttributeProto.name = 1, f = 2, i = 3, ints = 8; TensorProto.dims = 1,
data_type = 2, float_data = 4, name = 8, raw_data = 9).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

from .engine import NetConfig


class OnnxFormatError(ValueError):
    pass


# ---------------------------------------------------------------------------
# protobuf wire format
# ---------------------------------------------------------------------------
def _varint(b: bytes, i: int):
    shift = 0
    val = 0
    while True:
        c = b[i]
        i += 1
        val |= (c & 0x7F) << shift
        if not c & 0x80:
            return val, i
        shift += 7
        if shift > 70:
            raise OnnxFormatError("varint too long")


def _fields(b):
    """Yield (field_number, wire_type, value) for one message."""
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]
            i += 8
        elif wt == 2:
            ln, i = _varint(b, i)
            v = b[i:i + ln]
            i += ln
        elif wt == 5:
            v = b[i:i + 4]
            i += 4
        else:
            raise OnnxFormatError(f"unsupported wire type {wt}")
        yield fno, wt, v


def _packed_varints(v) -> list:
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(x)
    return out


def _sint64(x: int) -> int:
    return x - (1 << 64) if x >= (1 << 63) else x


@dataclass
class Node:
    op: str = ""
    name: str = ""
    inputs: list = field(default_factory=list)
    outputs: list = field(default_factory=list)
    attrs: dict = field(default_factory=dict)


def _parse_tensor(b) -> tuple:
    dims, dtype, name, raw, floats = [], 0, "", None, []
    for fno, wt, v in _fields(b):
        if fno == 1:
            dims += _packed_varints(v) if wt == 2 else [v]
        elif fno == 2:
            dtype = v
        elif fno == 8:
            name = bytes(v).decode()
        elif fno == 9:
            raw = bytes(v)
        elif fno == 4:
            if wt == 2:
                floats += list(struct.unpack(f"<{len(v) // 4}f", v))
            else:
                floats.append(struct.unpack("<f", v)[0])
        elif fno in (13, 14) and v:
            raise OnnxFormatError(f"initializer '{name}' uses external data, which is not supported")
    if dtype == 1:      # FLOAT
        arr = np.frombuffer(raw, dtype="<f4") if raw is not None else np.asarray(floats, dtype=np.float32)
    elif dtype == 7:    # INT64 (shape constants etc.)
        arr = np.frombuffer(raw, dtype="<i8") if raw is not None else np.zeros(0, np.int64)
    else:
        return name, None
    shape = [int(_sint64(d)) for d in dims]
    if shape and int(np.prod(shape)) != arr.size:
        raise OnnxFormatError(f"initializer '{name}': {arr.size} values for shape {shape}")
    return name, (arr.reshape(shape) if shape else arr)


def _parse_attr(b):
    name, val = "", None
    ints = []
    for fno, wt, v in _fields(b):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            val = struct.unpack("<f", v)[0]
        elif fno == 3:
            val = _sint64(v)
        elif fno == 4:
            val = bytes(v)
        elif fno == 5:                                   # AttributeProto.t: the tensor of a Constant node
            val = _parse_tensor(v)[1]
        elif fno == 8:
            ints += [_sint64(x) for x in _packed_varints(v)] if wt == 2 else [_sint64(v)]
    return name, (ints if ints else val)


def _parse_node(b) -> Node:
    n = Node()
    for fno, wt, v in _fields(b):
        if fno == 1:
            n.inputs.append(bytes(v).decode())
        elif fno == 2:
            n.outputs.append(bytes(v).decode())
        elif fno == 3:
            n.name = bytes(v).decode()
        elif fno == 4:
            n.op = bytes(v).decode()
        elif fno == 5:
            k, val = _parse_attr(v)
            n.attrs[k] = val
    return n


def _where(n: Node) -> str:
    return f"node '{n.name or (n.outputs[0] if n.outputs else '?')}' (op {n.op})"


def _parse_value_info_shape(b):
    """ValueInfoProto -> (name, [dims or None])."""
    name, dims = "", None
    for fno, wt, v in _fields(b):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:                                   # TypeProto
            for f2, _, v2 in _fields(v):
                if f2 == 1:                              # tensor_type
                    for f3, _, v3 in _fields(v2):
                        if f3 == 2:                      # shape
                            dims = []
                            for f4, _, v4 in _fields(v3):
                                if f4 == 1:              # dim
                                    dv = None
                                    for f5, w5, v5 in _fields(v4):
                                        if f5 == 1 and w5 == 0:
                                            dv = _sint64(v5)
                                    dims.append(dv)
    return name, dims


def parse_onnx(path_or_bytes):
    """-> (nodes in graph order, {initializer name: ndarray}, {graph input name: dims})."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        data = path_or_bytes
    else:
        with open(path_or_bytes, "rb") as fh:
            data = fh.read()
    data = memoryview(bytes(data))
    graph = None
    for fno, wt, v in _fields(data):
        if fno == 7 and wt == 2:
            graph = v
    if graph is None:
        raise OnnxFormatError("no GraphProto (field 7) in the file: not an ONNX model")
    nodes, inits, inputs = [], {}, {}
    for fno, wt, v in _fields(graph):
        if fno == 1:
            nodes.append(_parse_node(v))
        elif fno == 5:
            name, arr = _parse_tensor(v)
            if arr is not None:
                inits[name] = arr
        elif fno == 11:
            name, dims = _parse_value_info_shape(v)
            inputs[name] = dims
    return nodes, inits, inputs


# ---------------------------------------------------------------------------
# ConvTDFNet pattern -> engine tensors
# ---------------------------------------------------------------------------
def fold_constants(nodes, inits) -> dict:
    """Initialisers plus everything a chain of Constant / Identity / Transpose / Reshape / Squeeze / Unsqueeze / Cast nodes
    derives from them.  Exports made with ``do_constant_folding=False`` (or TrainingMode.PRESERVE) keep such nodes in
    the graph -- e.g. ``Transpose(linear.weight) -> MatMul`` -- where a folded export stores the result."""
    vals = dict(inits)
    for n in nodes:                                       # graph order is topological
        if not n.outputs:
            continue
        if n.op == "Constant":
            v = n.attrs.get("value")
            if isinstance(v, np.ndarray):
                vals[n.outputs[0]] = v
            continue
        if n.op not in ("Identity", "Transpose", "Reshape", "Squeeze", "Unsqueeze", "Cast") or not n.inputs or n.inputs[0] not in vals:
            continue
        x = vals[n.inputs[0]]
        if n.op in ("Identity", "Cast"):
            y = x
        elif n.op == "Transpose":
            perm = n.attrs.get("perm")
            y = np.transpose(x, perm if perm else None)
        elif n.op == "Reshape":
            if len(n.inputs) < 2 or n.inputs[1] not in vals:
                continue
            shape = [int(d) for d in np.asarray(vals[n.inputs[1]]).reshape(-1)]
            shape = [x.shape[i] if d == 0 else d for i, d in enumerate(shape)]
            y = x.reshape(shape)
        else:
            axes = n.attrs.get("axes")
            if axes is None and len(n.inputs) > 1 and n.inputs[1] in vals:
                axes = [int(a) for a in np.asarray(vals[n.inputs[1]]).reshape(-1)]
            if axes is None:
                continue
            axes = [axes] if isinstance(axes, int) else list(axes)
            y = x
            for ax in sorted(axes, reverse=(n.op == "Squeeze")):
                y = np.squeeze(y, ax) if n.op == "Squeeze" else np.expand_dims(y, ax)
        vals[n.outputs[0]] = y
    return vals


# ops of the ConvTDFNet eval graph (weights + structure), pass-through ops on activations, and ops that may only touch
# constants or SHAPE vectors (run-time Reshape targets)
_STRUCT_OPS = {"Conv", "ConvTranspose", "MatMul", "Gemm", "BatchNormalization", "InstanceNormalization", "Relu", "Add", "Mul"}
_PASS_OPS = {"Identity", "Dropout", "Transpose", "Reshape", "Flatten", "Cast"}
_SHAPE_OPS = {"Shape", "Gather", "Concat", "Slice", "Squeeze", "Unsqueeze", "Constant", "ConstantOfShape", "Sub", "Div", "Size", "Range"}


def check_ops(nodes, consts) -> None:
    """Every node must be an op this reader accounts for.  Ops that would change ACTIVATIONS and are not part of the net
    (Sigmoid, LeakyRelu, Pad, Resize ...) raise; so do shape-arithmetic ops fed with activations.  InstanceNormalization is accepted
    here and must then turn out to be the middle of torch's GroupNorm(2, c) lowering (convtdf_from_onnx: _gn_affine)."""
    shapeish = set(consts)                    # constants and everything computed from Shape(...) of an activation
    for n in nodes:
        if n.op == "Shape":
            shapeish.update(n.outputs)
            continue
        all_shape = bool(n.inputs) and all((i in shapeish or i == "") for i in n.inputs)
        if n.op in _SHAPE_OPS or (all_shape and n.op in _STRUCT_OPS | _PASS_OPS):
            if n.op in ("Constant", "ConstantOfShape") or all_shape:
                shapeish.update(n.outputs)
                continue
            if n.op in ("Squeeze", "Unsqueeze"):          # also legal on weights; never emitted on ConvTDFNet activations
                raise OnnxFormatError(f"{_where(n)}: acts on an activation; not part of ConvTDFNet")
            raise OnnxFormatError(f"{_where(n)}: shape arithmetic applied to an activation tensor; not part of ConvTDFNet")
        if n.op not in _STRUCT_OPS and n.op not in _PASS_OPS:
            raise OnnxFormatError(f"{_where(n)}: unsupported op -- this reader runs ConvTDFNet only "
                                  "(Conv, ConvTranspose, MatMul / Gemm, BatchNormalization or the GroupNorm lowering, Relu, Add, Mul, Transpose, Reshape)")


def _bn_affine(node: Node, inits):
    missing = [n for n in node.inputs[1:5] if n not in inits]
    if len(node.inputs) < 5 or missing:
        raise OnnxFormatError(f"{_where(node)}: scale / bias / mean / var must be constants (missing {missing})")
    g, b, m, v = (np.asarray(inits[n], np.float64) for n in node.inputs[1:5])
    eps = float(node.attrs.get("epsilon", 1e-5))
    scale = g / np.sqrt(v + eps)
    return scale, b - m * scale


def convtdf_from_onnx(path_or_bytes, dim_t: int | None = None):
    """Read an MDX-Net ``.onnx`` -> (NetConfig, {canonical name: float32 array}).

    ``dim_t`` overrides the time size found in the graph input (the net is
    convolutional in time; the reference re-exports through onnx2torch when
    segment_size != dim_t, mdx_separator.py:126-132)."""
    nodes, raw_inits, ginputs = parse_onnx(path_or_bytes)
    inits = fold_constants(nodes, raw_inits)
    check_ops(nodes, inits)
    consumers: dict = {}
    for idx, n in enumerate(nodes):
        for i in n.inputs:
            consumers.setdefault(i, []).append(idx)

    def next_op(idx, op):
        """The single consumer of node idx's FIRST output if it is `op` (else None); Reshape / Identity / Dropout / Flatten
        behind it are looked through (the Gemm lowering of a Linear puts a Reshape between the product and its BatchNorm)."""
        while True:
            outs = [j for j in consumers.get(nodes[idx].outputs[0], []) if nodes[j].op != "Shape" and (
                    (nodes[j].inputs and nodes[j].inputs[0] == nodes[idx].outputs[0]) or nodes[j].op in ("Add", "Mul"))]
            if len(outs) != 1:
                return None
            j = outs[0]
            if nodes[j].op == op:
                return j
            if nodes[j].op in ("Reshape", "Identity", "Dropout", "Flatten") and op not in ("Reshape", "Identity", "Dropout", "Flatten"):
                idx = j
                continue
            return None

    gn_used = set()

    def gn_affine(idx):
        """torch's lowering of GroupNorm(2, c) behind node idx (uvr_lib_v5/mdxnet.py:48-49, optimizer 'adamw'):
        Reshape([0, 2, -1]) -> InstanceNormalization(scale 1, bias 0) -> Reshape(Shape(x)) -> Mul(gamma [c,1,1]) -> Add(beta [c,1,1]).
        -> (gamma, beta) as float64 vectors, or None when no InstanceNormalization follows."""
        j = next_op(idx, "InstanceNormalization")
        if j is None:
            return None
        nd = nodes[j]
        rs = [k for k in range(len(nodes)) if nodes[k].outputs and nodes[k].outputs[0] == nd.inputs[0]]
        tgt = inits.get(nodes[rs[0]].inputs[1]) if rs and nodes[rs[0]].op == "Reshape" and len(nodes[rs[0]].inputs) > 1 else None
        if tgt is None or [int(v) for v in np.asarray(tgt).reshape(-1)][1:] != [2, -1]:
            raise OnnxFormatError(f"{_where(nd)}: InstanceNormalization that is not the GroupNorm(2, c) lowering (Reshape target {tgt})")
        sc, bi = (np.asarray(inits.get(n_), np.float64) if n_ in inits else None for n_ in nd.inputs[1:3])
        if sc is None or bi is None or sc.size != 2 or not np.all(sc == 1.0) or not np.all(bi == 0.0):
            raise OnnxFormatError(f"{_where(nd)}: GroupNorm lowering with a non-trivial per-group scale / bias")
        if abs(float(nd.attrs.get("epsilon", 1e-5)) - 1e-5) > 1e-9:
            raise OnnxFormatError(f"{_where(nd)}: GroupNorm epsilon {nd.attrs.get('epsilon')} (the engine uses nn.GroupNorm's 1e-5)")
        jm = next_op(j, "Mul")
        ja = next_op(jm, "Add") if jm is not None else None
        if jm is None or ja is None:
            raise OnnxFormatError(f"{_where(nd)}: GroupNorm lowering without its affine Mul / Add")
        gam = [i for i in nodes[jm].inputs if i in inits]
        bet = [i for i in nodes[ja].inputs if i in inits]
        if len(gam) != 1 or len(bet) != 1:
            raise OnnxFormatError(f"{_where(nd)}: GroupNorm affine operands are not constants")
        gn_used.add(j)
        return np.asarray(inits[gam[0]], np.float64).reshape(-1), np.asarray(inits[bet[0]], np.float64).reshape(-1)

    # ("conv", k, stride, w[cout,cin,kh,kw], b, gn) | ("convT", w[cin,cout,2,2], b, gn) | ("lin", w[n,k], bias|None, scale, shift, gn);
    # gn = (gamma, beta) of a GroupNorm(2, c) behind the layer (weights then unfolded, scale / shift None), else None
    layers = []
    norm_kind = []                                     # per layer: "bn" (BatchNorm folded), "gn" (GroupNorm behind it) or None (no norm at all)
    for idx, n in enumerate(nodes):
        if n.op == "Conv":
            w = np.asarray(inits[n.inputs[1]], np.float64)
            b = np.asarray(inits[n.inputs[2]], np.float64) if len(n.inputs) > 2 else np.zeros(w.shape[0])
            j = next_op(idx, "BatchNormalization")
            if j is not None:
                s, sh = _bn_affine(nodes[j], inits)
                w, b = w * s[:, None, None, None], b * s + sh
            gn = gn_affine(idx) if j is None else None
            ks = n.attrs.get("kernel_shape", list(w.shape[2:]))
            st = n.attrs.get("strides", [1, 1])
            if n.attrs.get("group", 1) != 1 or any(d != 1 for d in n.attrs.get("dilations", [1, 1])):
                raise OnnxFormatError(f"{_where(n)}: grouped / dilated Conv is not part of ConvTDFNet")
            if n.inputs[1] not in inits:
                raise OnnxFormatError(f"{_where(n)}: weight is not a constant")
            layers.append(("conv", int(ks[0]), int(st[0]), w, b, gn))
            norm_kind.append("bn" if j is not None else ("gn" if gn is not None else None))
        elif n.op == "ConvTranspose":
            w = np.asarray(inits[n.inputs[1]], np.float64)
            b = np.asarray(inits[n.inputs[2]], np.float64) if len(n.inputs) > 2 else np.zeros(w.shape[1])
            j = next_op(idx, "BatchNormalization")
            if j is not None:
                s, sh = _bn_affine(nodes[j], inits)
                w, b = w * s[None, :, None, None], b * s + sh
            layers.append(("convT", w, b, gn_affine(idx) if j is None else None))
            norm_kind.append("bn" if j is not None else ("gn" if layers[-1][-1] is not None else None))
        elif n.op == "MatMul":
            wname = n.inputs[1] if n.inputs[1] in inits else (n.inputs[0] if n.inputs[0] in inits else None)
            if wname is None:
                raise OnnxFormatError(f"{_where(n)}: MatMul without a constant operand")
            w = np.asarray(inits[wname], np.float64).T            # exporter stores W^T [in, out]
            bias = None
            j = idx
            ja = next_op(idx, "Add")
            if ja is not None and any(i in inits for i in nodes[ja].inputs):
                bias = np.asarray(inits[[i for i in nodes[ja].inputs if i in inits][0]], np.float64)
                j = ja
            jb = next_op(j, "BatchNormalization")
            if jb is None:
                gn = gn_affine(j)
                if gn is None:
                    raise OnnxFormatError(f"{_where(n)}: TDF Linear followed neither by BatchNormalization nor by the GroupNorm lowering")
                layers.append(("lin", w, bias, None, None, gn))
                norm_kind.append("gn")
                continue
            s, sh = _bn_affine(nodes[jb], inits)
            layers.append(("lin", w, bias, s, sh, None))
            norm_kind.append("bn")
        elif n.op == "Gemm":
            # Linear lowered as Reshape([-1, K]) -> Gemm(A, B, C) -> Reshape: Y = alpha * A * op(B) + beta * C
            if len(n.inputs) < 2 or n.inputs[1] not in inits or n.inputs[0] in inits:
                raise OnnxFormatError(f"{_where(n)}: expected activation x constant weight")
            if float(n.attrs.get("alpha", 1.0)) != 1.0 or float(n.attrs.get("beta", 1.0)) != 1.0 or int(n.attrs.get("transA", 0)) != 0:
                raise OnnxFormatError(f"{_where(n)}: alpha / beta / transA other than 1 / 1 / 0")
            w = np.asarray(inits[n.inputs[1]], np.float64)
            if not int(n.attrs.get("transB", 0)):
                w = w.T                                            # B is [K, N]: Linear's weight is its transpose
            bias = None
            if len(n.inputs) > 2 and n.inputs[2]:
                if n.inputs[2] not in inits:
                    raise OnnxFormatError(f"{_where(n)}: bias operand is not a constant")
                bias = np.asarray(inits[n.inputs[2]], np.float64).reshape(-1)
            jb = next_op(idx, "BatchNormalization")
            if jb is None:
                gn = gn_affine(idx)
                if gn is None:
                    raise OnnxFormatError(f"{_where(n)}: TDF Linear followed neither by BatchNormalization nor by the GroupNorm lowering")
                layers.append(("lin", w, bias, None, None, gn))
                norm_kind.append("gn")
                continue
            s, sh = _bn_affine(nodes[jb], inits)
            layers.append(("lin", w, bias, s, sh, None))
            norm_kind.append("bn")

    if len(layers) < 4 or layers[0][0] != "conv" or layers[0][1] != 1:
        raise OnnxFormatError("graph does not start with the 1x1 first_conv of ConvTDFNet")
    stray = [nd for j, nd in enumerate(nodes) if nd.op == "InstanceNormalization" and j not in gn_used]
    if stray:
        raise OnnxFormatError(f"{_where(stray[0])}: InstanceNormalization outside the GroupNorm(2, c) lowering of a conv / linear"
                              + (f" (and {len(stray) - 1} more)" if len(stray) > 1 else ""))
    # Every layer but the final conv carries a norm, all of one kind.  A conv WITHOUT a norm node counts as BatchNorm: torch's exporter folds an
    # eval-mode BatchNorm into the Conv / ConvTranspose in front of it, so a folded norm and no norm at all are the same graph (ADVICE r5 asked for
    # norm-less layers to be rejected: for convs they cannot be told apart; a Linear without either norm IS rejected above).  What can be
    # checked: GroupNorm and BatchNorm layers never mix -- a GroupNorm graph has the lowering behind EVERY layer.
    kinds = {"gn" if k == "gn" else "bn" for k in norm_kind[:-1]}
    if len(kinds) != 1:
        first_gn = norm_kind.index("gn")
        raise OnnxFormatError(f"graph mixes GroupNorm layers (first: layer {first_gn}) with BatchNorm / norm-less ones; not a ConvTDFNet")
    group = kinds.pop() == "gn"
    out: dict = {}
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731

    def put_gn(dst, L):
        if L[-1] is not None:
            out[f"{dst}.gn_w"], out[f"{dst}.gn_b"] = f32(L[-1][0]), f32(L[-1][1])
    first = layers[0]
    g, dim_c = first[3].shape[0], first[3].shape[1]
    out["first.w"], out["first.b"] = f32(first[3].reshape(g, dim_c)), f32(first[4])
    put_gn("first", first)
    pos = 1

    def take_block(dst):
        nonlocal pos
        l = 0
        while pos < len(layers) and layers[pos][0] == "conv" and layers[pos][1] == 3:
            out[f"{dst}.tfc{l}.w"], out[f"{dst}.tfc{l}.b"] = f32(layers[pos][3]), f32(layers[pos][4])
            put_gn(f"{dst}.tfc{l}", layers[pos])
            l += 1
            pos += 1
        dims = []
        # TDF branch (modules.py:52-70): two linears (bn > 0), ONE Linear(f, f) (bn == 0), or none (bn is None)
        for t in range(2):
            if pos >= len(layers) or layers[pos][0] != "lin":
                break
            _, w, bias, s, sh, _gn = layers[pos]
            out[f"{dst}.tdf{t}.w"] = f32(w)
            if bias is not None:
                out[f"{dst}.tdf{t}.bias"] = f32(bias)
            if s is not None:
                out[f"{dst}.tdf{t}.scale"], out[f"{dst}.tdf{t}.shift"] = f32(s), f32(sh)
            put_gn(f"{dst}.tdf{t}", layers[pos])
            dims.append(w.shape)
            pos += 1
        if len(dims) == 1 and dims[0][0] != dims[0][1]:
            raise OnnxFormatError(f"{dst}: a single TDF linear must be square (bn == 0), got {dims[0]}")
        return l, dims

    l0, dims0 = take_block("enc0")
    if dims0:
        dim_f = dims0[0][1]
        bn = 0 if len(dims0) == 1 else dim_f // dims0[0][0]
    else:                                              # no TDF branch: the frequency size is the graph input's
        bn = None
        dim_f = next((int(dims[2]) for name, dims in ginputs.items() if name not in raw_inits and dims and len(dims) == 4 and dims[2]), None)
        if dim_f is None:
            raise OnnxFormatError("ConvTDFNet without a TDF branch and without a recorded input shape: dim_f unknown")
    has_bias = "enc0.tdf0.bias" in out
    # encoder: blocks separated by 2x2 stride-2 convs
    enc_blocks = 1
    n = 0
    while pos < len(layers) and layers[pos][0] == "conv" and layers[pos][1] == 2 and layers[pos][2] == 2:
        out[f"ds{n}.w"], out[f"ds{n}.b"] = f32(layers[pos][3]), f32(layers[pos][4])
        put_gn(f"ds{n}", layers[pos])
        pos += 1
        n += 1
        name = f"enc{enc_blocks}"
        take_block(name)
        enc_blocks += 1
    # the block after the last ds is the bottleneck: rename enc<n> -> mid
    for k in [k for k in out if k.startswith(f"enc{n}.")]:
        out["mid." + k.split(".", 1)[1]] = out.pop(k)
    for i in range(n):
        if pos >= len(layers) or layers[pos][0] != "convT":
            raise OnnxFormatError(f"expected ConvTranspose us{i}")
        out[f"us{i}.w"], out[f"us{i}.b"] = f32(layers[pos][1]), f32(layers[pos][2])
        put_gn(f"us{i}", layers[pos])
        pos += 1
        take_block(f"dec{i}")
    if pos != len(layers) - 1 or layers[pos][0] != "conv" or layers[pos][1] != 1:
        raise OnnxFormatError("graph does not end with the 1x1 final_conv of ConvTDFNet")
    fw = layers[pos][3]
    out["final.w"], out["final.b"] = f32(fw.reshape(fw.shape[0], fw.shape[1])), f32(layers[pos][4])

    if dim_t is None:
        for name, dims in ginputs.items():
            if name not in raw_inits and dims and len(dims) == 4 and dims[3]:
                dim_t = int(dims[3])
        if dim_t is None:
            raise OnnxFormatError("time size not recorded in the graph input; pass dim_t")
    cfg = NetConfig(dim_c=int(dim_c), dim_f=int(dim_f), dim_t=int(dim_t), g=int(g), l=int(l0),
                    num_blocks=2 * n + 1, k=3, bn=(None if bn is None else int(bn)), tdf_bias=bool(has_bias), norm="group" if group else "batch")
    return cfg, out
