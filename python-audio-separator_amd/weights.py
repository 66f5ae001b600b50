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


def _raw_conv(sd, conv, norm, dst, out, transposed=False):
    """GroupNorm variant: the conv goes over unfolded, the norm's affine beside it (<dst>.gn_w / .gn_b)."""
    w = _np64(sd[conv + ".weight"])
    cout = w.shape[1] if transposed else w.shape[0]
    b = _np64(sd[conv + ".bias"]) if (conv + ".bias") in sd else np.zeros(cout)
    out[f"{dst}.w"], out[f"{dst}.b"] = w.astype(np.float32), b.astype(np.float32)
    out[f"{dst}.gn_w"] = _np64(sd[norm + ".weight"]).astype(np.float32)
    out[f"{dst}.gn_b"] = _np64(sd[norm + ".bias"]).astype(np.float32)


def _fold_block(sd, src, dst, l, out, tdf_bias, group=False):
    for j in range(l):
        if group:
            _raw_conv(sd, f"{src}.tfc.H.{j}.0", f"{src}.tfc.H.{j}.1", f"{dst}.tfc{j}", out)
        else:
            w, b = _fold_conv(sd, f"{src}.tfc.H.{j}.0", f"{src}.tfc.H.{j}.1")
            out[f"{dst}.tfc{j}.w"], out[f"{dst}.tfc{j}.b"] = w, b
    # bn > 0: tdf.0 / tdf.1 / (ReLU) / tdf.3 / tdf.4; bn == 0: tdf.0 / tdf.1 only; bn is None: no tdf module (modules.py:52-70)
    for idx, (lin, bn) in enumerate((("tdf.0", "tdf.1"), ("tdf.3", "tdf.4"))):
        if f"{src}.{lin}.weight" not in sd:
            continue
        out[f"{dst}.tdf{idx}.w"] = _np64(sd[f"{src}.{lin}.weight"]).astype(np.float32)
        if tdf_bias:
            out[f"{dst}.tdf{idx}.bias"] = _np64(sd[f"{src}.{lin}.bias"]).astype(np.float32)
        if group:
            out[f"{dst}.tdf{idx}.gn_w"] = _np64(sd[f"{src}.{bn}.weight"]).astype(np.float32)
            out[f"{dst}.tdf{idx}.gn_b"] = _np64(sd[f"{src}.{bn}.bias"]).astype(np.float32)
        else:
            scale, shift = _bn_affine(sd, f"{src}.{bn}")
            out[f"{dst}.tdf{idx}.scale"] = scale.astype(np.float32)
            out[f"{dst}.tdf{idx}.shift"] = shift.astype(np.float32)


def state_norm_kind(sd: dict) -> str:
    """'batch' when the state_dict's norms carry running statistics (BatchNorm2d: optimizer 'rmsprop'), 'group' when they are
    affine only (GroupNorm(2, c): optimizer 'adamw', uvr_lib_v5/mdxnet.py:45-49)."""
    return "batch" if "first_conv.1.running_mean" in sd else "group"


def fold_convtdf_state(sd: dict, num_blocks: int, l: int, tdf_bias: bool = False, norm: str | None = None) -> dict:
    """norm: 'batch' / 'group' / None = read it off the state_dict (state_norm_kind)."""
    norm = norm or state_norm_kind(sd)
    if norm == "group":
        return _group_convtdf_state(sd, num_blocks, l, tdf_bias)
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


def _group_convtdf_state(sd: dict, num_blocks: int, l: int, tdf_bias: bool) -> dict:
    """The GroupNorm(2, c) variant: nothing can be folded (the statistics depend on the input), every tensor goes over as it is."""
    n = num_blocks // 2
    out: dict = {}
    _raw_conv(sd, "first_conv.0", "first_conv.1", "first", out)
    out["first.w"] = out["first.w"].reshape(out["first.w"].shape[0], out["first.w"].shape[1])
    for i in range(n):
        _fold_block(sd, f"encoding_blocks.{i}", f"enc{i}", l, out, tdf_bias, group=True)
        _raw_conv(sd, f"ds.{i}.0", f"ds.{i}.1", f"ds{i}", out)
    _fold_block(sd, "bottleneck_block", "mid", l, out, tdf_bias, group=True)
    for i in range(n):
        _raw_conv(sd, f"us.{i}.0", f"us.{i}.1", f"us{i}", out, transposed=True)
        _fold_block(sd, f"decoding_blocks.{i}", f"dec{i}", l, out, tdf_bias, group=True)
    wf = _np64(sd["final_conv.0.weight"])
    out["final.w"] = wf.reshape(wf.shape[0], wf.shape[1]).astype(np.float32)
    out["final.b"] = _np64(sd["final_conv.0.bias"]).astype(np.float32)
    return out
