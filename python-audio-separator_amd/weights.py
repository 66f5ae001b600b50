"""Weight container -> engine tensors.

Takes a ConvTDFNet ``state_dict`` with the reference's parameter names
(uvr_lib_v5/mdxnet.py:54-95, modules.py) -- torch tensors or numpy arrays --
folds every eval-mode BatchNorm2d into the preceding conv / linear in float64,
and returns the canonical named float32 tensors that ``asx_net_set_tensor``
expects (include/asx.h).
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5


def _np64(t) -> np.ndarray:
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.asarray(t, dtype=np.float64)


def _bn_affine(sd, prefix):
    g, b = _np64(sd[prefix + ".weight"]), _np64(sd[prefix + ".bias"])
    m, v = _np64(sd[prefix + ".running_mean"]), _np64(sd[prefix + ".running_var"])
    scale = g / np.sqrt(v + BN_EPS)
    return scale, b - m * scale


def _fold_conv(sd, conv, bn, transposed=False):
    """conv + BN -> (w', b').  transposed: ConvTranspose2d weight [cin, cout, kh, kw]."""
    w = _np64(sd[conv + ".weight"])
    cout = w.shape[1] if transposed else w.shape[0]
    b = _np64(sd[conv + ".bias"]) if (conv + ".bias") in sd else np.zeros(cout)
    scale, shift = _bn_affine(sd, bn)
    if transposed:
        w = w * scale[None, :, None, None]
    else:
        w = w * scale[:, None, None, None]
    return w.astype(np.float32), (b * scale + shift).astype(np.float32)


def _fold_block(sd, src, dst, l, out, tdf_bias):
    for j in range(l):
        w, b = _fold_conv(sd, f"{src}.tfc.H.{j}.0", f"{src}.tfc.H.{j}.1")
        out[f"{dst}.tfc{j}.w"], out[f"{dst}.tfc{j}.b"] = w, b
    for idx, (lin, bn) in enumerate((("tdf.0", "tdf.1"), ("tdf.3", "tdf.4"))):
        out[f"{dst}.tdf{idx}.w"] = _np64(sd[f"{src}.{lin}.weight"]).astype(np.float32)
        if tdf_bias:
            out[f"{dst}.tdf{idx}.bias"] = _np64(sd[f"{src}.{lin}.bias"]).astype(np.float32)
        scale, shift = _bn_affine(sd, f"{src}.{bn}")
        out[f"{dst}.tdf{idx}.scale"] = scale.astype(np.float32)
        out[f"{dst}.tdf{idx}.shift"] = shift.astype(np.float32)


def fold_convtdf_state(sd: dict, num_blocks: int, l: int, tdf_bias: bool = False) -> dict:
    n = num_blocks // 2
    out: dict = {}
    w, b = _fold_conv(sd, "first_conv.0", "first_conv.1")
    out["first.w"], out["first.b"] = w.reshape(w.shape[0], w.shape[1]), b
    for i in range(n):
        _fold_block(sd, f"encoding_blocks.{i}", f"enc{i}", l, out, tdf_bias)
        out[f"ds{i}.w"], out[f"ds{i}.b"] = _fold_conv(sd, f"ds.{i}.0", f"ds.{i}.1")
    _fold_block(sd, "bottleneck_block", "mid", l, out, tdf_bias)
    for i in range(n):
        out[f"us{i}.w"], out[f"us{i}.b"] = _fold_conv(sd, f"us.{i}.0", f"us.{i}.1", transposed=True)
        _fold_block(sd, f"decoding_blocks.{i}", f"dec{i}", l, out, tdf_bias)
    wf = _np64(sd["final_conv.0.weight"])
    out["final.w"] = wf.reshape(wf.shape[0], wf.shape[1]).astype(np.float32)
    out["final.b"] = _np64(sd["final_conv.0.bias"]).astype(np.float32)
    return out
