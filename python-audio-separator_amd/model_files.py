"""Model files of the Demucs and VR plugins: what the reference reads with ``torch.load`` + its own classes.

Demucs (reference: uvr_lib_v5/demucs/pretrained.py:get_model :62-81, repo.py LocalRepo :73-108 / BagOnlyRepo :111-142 /
AnyModelRepo :145-158, states.py:load_model :34-60, apply.py:BagOfModels :26-66): a model *name* resolves, inside the
directory of ``model_path``, either to ``<sig>.th`` / ``<sig>-<checksum>.th`` (one model) or to ``<name>.yaml``
(a bag: ``models`` = signatures, optional per-source ``weights`` and ``segment``).  A ``.th`` file is a pickled package
``{klass, args, kwargs, state}``; the reference unpickles it with the vendored ``demucs`` package on sys.path so that
``klass`` resolves to its HTDemucs / HDemucs class.  Here no Demucs module is ever imported or executed: the pickle is
read by torch's restricted (``weights_only``) unpickler with the class references mapped to inert markers, and the
constructor arguments become an engine configuration (HTConfig / HDConfig).

VR (reference: architectures/vr_separator.py:158-176): a bare ``state_dict`` saved with ``torch.save``.
"""
from __future__ import annotations

import fractions
import hashlib
import os

# positional order of the constructors (htdemucs.py:56-133, hdemucs.py:362-410): package["args"] binds by position
HT_PARAMS = ["sources", "audio_channels", "channels", "channels_time", "growth", "nfft", "wiener_iters", "end_iters",
             "wiener_residual", "cac", "depth", "rewrite", "multi_freqs", "multi_freqs_depth", "freq_emb", "emb_scale", "emb_smooth",
             "kernel_size", "time_stride", "stride", "context", "context_enc", "norm_starts", "norm_groups", "dconv_mode",
             "dconv_depth", "dconv_comp", "dconv_init", "bottom_channels", "t_layers", "t_emb", "t_hidden_scale", "t_heads",
             "t_dropout", "t_max_positions", "t_norm_in", "t_norm_in_group", "t_group_norm", "t_norm_first", "t_norm_out",
             "t_max_period", "t_weight_decay", "t_lr", "t_layer_scale", "t_gelu", "t_weight_pos_embed", "t_sin_random_shift",
             "t_cape_mean_normalize", "t_cape_augment", "t_cape_glob_loc_scale", "t_sparse_self_attn", "t_sparse_cross_attn",
             "t_mask_type", "t_mask_random_seed", "t_sparse_attn_window", "t_global_window", "t_sparsity", "t_auto_sparsity",
             "t_cross_first", "rescale", "samplerate", "segment", "use_train_segment"]
HD_PARAMS = ["sources", "audio_channels", "channels", "channels_time", "growth", "nfft", "wiener_iters", "end_iters",
             "wiener_residual", "cac", "depth", "rewrite", "hybrid", "hybrid_old", "multi_freqs", "multi_freqs_depth", "freq_emb",
             "emb_scale", "emb_smooth", "kernel_size", "time_stride", "stride", "context", "context_enc", "norm_starts",
             "norm_groups", "dconv_mode", "dconv_depth", "dconv_comp", "dconv_attn", "dconv_lstm", "dconv_init", "rescale",
             "samplerate", "segment"]

_KLASS_MODULES = ("demucs.", "audio_separator.separator.uvr_lib_v5.demucs.")
_KLASS_NAMES = {"htdemucs.HTDemucs": "HTDemucs", "hdemucs.HDemucs": "HDemucs", "demucs.Demucs": "Demucs"}


class ModelLoadingError(RuntimeError):
    """repo.py:ModelLoadingError."""


def _klass_markers():
    """Inert stand-ins for the classes a package's ``klass`` may reference; their (module, qualname) is what the
    restricted unpickler matches the pickle's GLOBAL opcode against."""
    out = []
    for prefix in _KLASS_MODULES:
        for tail, kind in _KLASS_NAMES.items():
            mod, name = tail.split(".")
            marker = type(name, (), {"asx_kind": kind})
            marker.__module__ = prefix + mod
            marker.__qualname__ = name
            out.append(marker)
    return out


def read_demucs_package(path: str) -> dict:
    """states.py:load_model without executing model code: {"kind", "kwargs", "state"}.

    ``kind`` is "HTDemucs" / "HDemucs" / "Demucs"; ``kwargs`` is args bound by position + kwargs (unknown names dropped
    like the non-strict branch, states.py:50-57)."""
    import torch
    allowed = _klass_markers() + [fractions.Fraction]
    try:
        with torch.serialization.safe_globals(allowed):
            package = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:
        raise ModelLoadingError(f"{path}: not a Demucs package this loader accepts ({e})") from e
    if not isinstance(package, dict) or not {"klass", "args", "kwargs", "state"} <= set(package):
        raise ModelLoadingError(f"{path}: expected a package with klass / args / kwargs / state")
    kind = getattr(package["klass"], "asx_kind", None)
    if kind is None:
        raise ModelLoadingError(f"{path}: unknown model class {package['klass']!r}")
    names = {"HTDemucs": HT_PARAMS, "HDemucs": HD_PARAMS}.get(kind)
    if names is None:
        raise NotImplementedError(f"{path}: the waveform-only Demucs v1 / v2 class is not accelerated")
    kwargs = dict(zip(names, package["args"]))
    kwargs.update({k: v for k, v in package["kwargs"].items() if k in names})
    state = package["state"]
    if isinstance(state, dict) and state.get("__quantized"):
        raise NotImplementedError(f"{path}: diffq-quantised checkpoints are not supported")
    return {"kind": kind, "kwargs": kwargs, "state": state, "path": path}


def _check_checksum(path: str, checksum: str):
    """repo.py:check_checksum: sha256 prefix encoded in the file name ``<sig>-<checksum>.th``."""
    sha = hashlib.sha256()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            sha.update(block)
    actual = sha.hexdigest()[: len(checksum)]
    if actual != checksum:
        raise ModelLoadingError(f"Invalid checksum for file {path}, expected {checksum} but got {actual}")


def get_demucs_model(name: str, repo: str) -> dict:
    """pretrained.get_model(name, repo=Path(dir)) (pretrained.py:62-81): {"models": [package...], "weights", "segment",
    "is_bag"}.  A single ``.th`` is returned as a one-model non-bag (demucs_segments then never changes its segment,
    apply.py:263-300)."""
    import yaml
    if not os.path.isdir(repo):
        raise ModelLoadingError(f"{repo} must exist and be a directory.")
    models, checksums, bags = {}, {}, {}
    for fname in sorted(os.listdir(repo)):
        stem, ext = os.path.splitext(fname)
        if ext == ".th":
            sig = stem
            if "-" in stem:
                sig, checksum = stem.split("-")
                checksums[sig] = checksum
            if sig in models:
                raise ModelLoadingError(f"Duplicate pre-trained model exist for signature {sig}. Please delete all but one.")
            models[sig] = os.path.join(repo, fname)
        elif ext == ".yaml":
            bags[stem] = os.path.join(repo, fname)

    def load_sig(sig):
        if sig not in models:
            raise ModelLoadingError(f"Could not find pre-trained model with signature {sig}.")
        if sig in checksums:
            _check_checksum(models[sig], checksums[sig])
        return read_demucs_package(models[sig])

    if name in models:
        return {"models": [load_sig(name)], "weights": None, "segment": None, "is_bag": False}
    if name not in bags:
        raise ModelLoadingError(f"{name} is neither a single pre-trained model or a bag of models.")
    with open(bags[name]) as f:
        bag = yaml.safe_load(f)
    packages = [load_sig(sig) for sig in bag["models"]]
    first = packages[0]["kwargs"]
    for p in packages[1:]:
        for key, default in (("sources", None), ("samplerate", 44100), ("audio_channels", 2)):
            if p["kwargs"].get(key, default) != first.get(key, default):
                raise ModelLoadingError(f"bag {name}: models disagree on {key}")
    weights = bag.get("weights")
    if weights is not None:
        if len(weights) != len(packages) or any(len(w) != len(first["sources"]) for w in weights):
            raise ModelLoadingError(f"bag {name}: weights must be one list of {len(first['sources'])} values per model")
    return {"models": packages, "weights": weights, "segment": bag.get("segment"), "is_bag": True}


def read_state_dict(path: str) -> dict:
    """torch.load(model_path, map_location="cpu") of a bare state_dict (vr_separator.py:176, mdxc_separator.py:109)."""
    sd = safe_torch_load(path)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return sd


UNSAFE_ENV = "ASX_ALLOW_UNSAFE_PICKLE"


def safe_torch_load(path: str):
    """torch.load restricted to tensors and plain containers (weights_only=True -- what the reference's plain torch.load does
    on torch >= 2.6).  A file that needs arbitrary pickle globals is REFUSED: un-pickling it would run code from a downloaded
    model.  Setting ASX_ALLOW_UNSAFE_PICKLE=1 opts into the unrestricted loader for a file the user trusts (logged)."""
    import logging
    import pickle
    import torch
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        # only a REFUSAL (a global outside the allow-list) is answered with the opt-out hint; a missing, truncated or corrupt
        # file (OSError, EOFError, zipfile.BadZipFile, RuntimeError from the zip reader ...) propagates as what it is
        if "weights_only" not in str(e).lower() and "unsupported global" not in str(e).lower() and "unsupported class" not in str(e).lower() \
                and "unsupported operand" not in str(e).lower():
            raise
        if os.environ.get(UNSAFE_ENV, "") != "1":
            raise ModelLoadingError(
                f"{path}: refused by the restricted (weights_only) loader: {e}.  If you trust this file, set {UNSAFE_ENV}=1 "
                "to un-pickle it without restrictions.") from e
        logging.getLogger(__name__).warning("%s: loading with UNRESTRICTED pickle (%s=1) -- the file can execute code", path, UNSAFE_ENV)
        return torch.load(path, map_location="cpu", weights_only=False)
