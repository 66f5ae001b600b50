"""Roformer model-file loading for the MDXC plugin: checkpoint + YAML -> (RofConfig, state_dict).

Mirror of the reference's loader stack for this path -- audio_separator/separator/roformer/roformer_loader.py
(load_model :23-66, _create_bs_roformer :123-150, _create_mel_band_roformer :152-195, legacy fallback :197-236),
configuration_normalizer.py (structure flattening :74-100, aliases :102-146, value coercion :160-224, path-based type
detection :266-300) and parameter_validator.py / *_validator.py defaults (:297-337, bs :199-221, mel :268-288) -- but
instead of instantiating a torch module it produces the constructor arguments the reference would have used, which
the engine turns into its own layout (engine.RofConfig, asx_rof_config).  The quirks are kept on purpose, because
they decide what the reference computes: BS-Roformer never receives ``mask_estimator_depth`` from the YAML (the
class default 2 is used), a YAML without ``stft_hop_length`` gets the validator's 512, ``training`` / ``inference``
sections only contribute dim_t / hop_length / n_fft / sample_rate, ``audio`` is not flattened.
"""
from __future__ import annotations

import copy
import logging
import os

logger = logging.getLogger(__name__)

_ALIASES = {
    "n_fft": "stft_n_fft", "hop_length": "stft_hop_length", "win_length": "stft_win_length", "window_fn": "stft_window_fn",
    "normalized": "stft_normalized", "n_heads": "heads", "num_heads": "heads", "head_dim": "dim_head", "dropout": "attn_dropout",
    "attention_dropout": "attn_dropout", "feedforward_dropout": "ff_dropout", "expansion_factor": "mlp_expansion_factor",
    "mlp_ratio": "mlp_expansion_factor", "use_checkpoint": "use_torch_checkpoint", "checkpoint": "use_torch_checkpoint",
    "freq_bands": "freqs_per_bands", "frequency_bands": "freqs_per_bands", "mel_bands": "num_bands", "n_mels": "num_bands",
}
_BOOL_KEYS = {"stereo", "flash_attn", "sage_attention", "zero_dc", "use_torch_checkpoint", "skip_connection", "stft_normalized"}
_INT_KEYS = {"dim", "depth", "num_stems", "time_transformer_depth", "freq_transformer_depth", "dim_head", "heads",
             "mlp_expansion_factor", "num_bands", "sample_rate", "stft_n_fft", "stft_hop_length", "stft_win_length",
             "mask_estimator_depth"}
_FLOAT_KEYS = {"attn_dropout", "ff_dropout", "multi_stft_resolution_loss_weight", "fmin", "fmax"}

# bs_roformer.py:228-297 DEFAULT_FREQS_PER_BANDS: 24 x 2, 12 x 4, 8 x 12, 8 x 24, 8 x 48, 128, 129 (sums to 1025)
DEFAULT_FREQS_PER_BANDS = (2,) * 24 + (4,) * 12 + (12,) * 8 + (24,) * 8 + (48,) * 8 + (128, 129)

_BASE_DEFAULTS = {"stereo": False, "num_stems": 2, "time_transformer_depth": 2, "freq_transformer_depth": 2, "dim_head": 64,
                  "heads": 8, "attn_dropout": 0.0, "ff_dropout": 0.0, "flash_attn": True, "mlp_expansion_factor": 4,
                  "sage_attention": False, "zero_dc": True, "use_torch_checkpoint": False, "skip_connection": False,
                  "sample_rate": 44100, "norm": None}


class ParameterValidationError(ValueError):
    """roformer/parameter_validation_error.py: raised for a missing / mistyped required parameter."""


def detect_model_type(config: dict):
    """configuration_normalizer.py:237-264."""
    if "freqs_per_bands" in config:
        return "bs_roformer"
    if "num_bands" in config or "n_mels" in config or "mel_bands" in config:
        return "mel_band_roformer"
    hint = config.get("model_type", config.get("type", config.get("architecture")))
    if isinstance(hint, str):
        h = hint.lower()
        if "bs" in h and "roformer" in h:
            return "bs_roformer"
        if "mel" in h and "roformer" in h:
            return "mel_band_roformer"
        if "roformer" in h:
            return "bs_roformer"
    return None


def model_type_from_path(config: dict, file_path: str) -> str:
    """configuration_normalizer.py:283-298: the file name wins over the configuration."""
    p = file_path.lower()
    if "bs" in p and "roformer" in p:
        return "bs_roformer"
    if "mel" in p and "roformer" in p:
        return "mel_band_roformer"
    return detect_model_type(config) or "bs_roformer"


def _coerce(key, value):
    if key in _BOOL_KEYS:
        return value.lower() in ("true", "1", "yes", "on") if isinstance(value, str) else bool(value)
    if key in _INT_KEYS:
        if isinstance(value, str):
            try:
                return int(float(value))
            except (ValueError, TypeError):
                return value
        return int(value) if isinstance(value, (int, float)) else value
    if key in _FLOAT_KEYS:
        if isinstance(value, str):
            try:
                return float(value)
            except (ValueError, TypeError):
                return value
        return float(value) if isinstance(value, (int, float)) else value
    if key.startswith("freqs_per_bands"):
        if isinstance(value, str):
            body = value.strip("()[]").replace(" ", "")
            try:
                return tuple(int(x) for x in body.split(",")) if body else value
            except (ValueError, TypeError):
                return value
        return tuple(value) if isinstance(value, list) else value
    if key in ("norm", "act", "mel_scale"):
        return str(value).lower() if value is not None else value
    return value


def defaults_for(model_type: str) -> dict:
    d = dict(_BASE_DEFAULTS)
    if model_type == "bs_roformer":
        d.update(freqs_per_bands=(2, 4, 8, 16, 32, 64), mask_estimator_depth=2,   # bs_roformer_validator.py:19,213
                 stft_n_fft=2048, stft_hop_length=512,
                 stft_win_length=2048, multi_stft_resolution_loss_weight=1.0)
    elif model_type == "mel_band_roformer":
        d.update(num_bands=64, fmin=0, fmax=None, mel_scale="htk")
    return d


def normalize_config(config: dict, model_type: str, apply_defaults: bool = True, validate: bool = True) -> dict:
    """configuration_normalizer.py:31-71: flatten -> rename -> coerce -> defaults -> validate."""
    flat: dict = {}
    for key, value in copy.deepcopy(config).items():
        if isinstance(value, dict) and key in ("model", "architecture", "params"):
            flat.update(value)
        elif key in ("training", "inference") and isinstance(value, dict):
            for k, v in value.items():
                if k in ("dim_t", "hop_length", "n_fft", "sample_rate"):
                    flat[k] = v
        else:
            flat[key] = value
    renamed = {_ALIASES.get(k, k): v for k, v in flat.items()}
    out = {k: _coerce(k, v) for k, v in renamed.items()}
    if apply_defaults:
        merged = defaults_for(model_type)
        merged.update(out)
        out = merged
    if validate:
        _validate(out, model_type)
    return out


def _validate(cfg: dict, model_type: str):
    """The checks of parameter_validator.py / bs_roformer_validator.py / mel_band_roformer_validator.py that can fail a load:
    required keys present and of integer type, band description present."""
    required = ["dim", "depth"] + (["freqs_per_bands"] if model_type == "bs_roformer" else ["num_bands"])
    for key in required:
        if key not in cfg:
            raise ParameterValidationError(f"Missing required parameter '{key}' for {model_type}")
    for key in ("dim", "depth"):
        if not isinstance(cfg[key], int) or isinstance(cfg[key], bool) or cfg[key] <= 0:
            raise ParameterValidationError(f"Parameter '{key}' must be a positive int, got {cfg[key]!r}")
    if model_type == "bs_roformer":
        f = cfg["freqs_per_bands"]
        if not isinstance(f, (tuple, list)) or len(f) < 2 or not all(isinstance(x, int) and x > 0 for x in f):
            raise ParameterValidationError(f"Parameter 'freqs_per_bands' must be a tuple of positive ints, got {f!r}")
    elif not isinstance(cfg["num_bands"], int) or cfg["num_bands"] <= 0:
        raise ParameterValidationError(f"Parameter 'num_bands' must be a positive int, got {cfg['num_bands']!r}")


def constructor_args(cfg: dict, model_type: str) -> dict:
    """What _create_bs_roformer / _create_mel_band_roformer pass to the class (roformer_loader.py:123-195), completed with
    the class defaults of the arguments they do not pass (bs_roformer.py:303-340, mel_band_roformer.py:196-230)."""
    a = {"dim": cfg["dim"], "depth": cfg["depth"], "stereo": cfg.get("stereo", False), "num_stems": cfg.get("num_stems", 2),
         "time_transformer_depth": cfg.get("time_transformer_depth", 2), "freq_transformer_depth": cfg.get("freq_transformer_depth", 2),
         "dim_head": cfg.get("dim_head", 64), "heads": cfg.get("heads", 8), "mlp_expansion_factor": cfg.get("mlp_expansion_factor", 4),
         "stft_n_fft": 2048, "stft_hop_length": 512, "stft_win_length": 2048, "stft_normalized": False, "linear_transformer_depth": 0}
    for k in ("stft_n_fft", "stft_hop_length", "stft_win_length"):
        if k in cfg:
            a[k] = cfg[k]
    if model_type == "bs_roformer":
        a["freqs_per_bands"] = tuple(cfg["freqs_per_bands"])
        a["mask_estimator_depth"] = 2                       # never forwarded from the configuration (roformer_loader.py:125-149)
    else:
        a["num_bands"] = cfg["num_bands"]
        a["sample_rate"] = cfg.get("sample_rate", 44100)
        a["mask_estimator_depth"] = cfg.get("mask_estimator_depth", 1)
        if "stft_normalized" in cfg:
            a["stft_normalized"] = cfg["stft_normalized"]
        if cfg.get("stft_window_fn") is not None:           # forwarded for the mel class only (roformer_loader.py:175-191)
            a["stft_window_fn"] = cfg["stft_window_fn"]
    return a


def legacy_constructor_args(model_cfg: dict, model_type: str) -> dict:
    """``BSRoformer(**model_cfg)`` / ``MelBandRoformer(**model_cfg)`` of the legacy path (roformer_loader.py:209-217): every
    key of the YAML's model section reaches the class, the rest are the class defaults."""
    a = {"stereo": False, "num_stems": 1, "time_transformer_depth": 2, "freq_transformer_depth": 2, "dim_head": 64, "heads": 8,
         "mlp_expansion_factor": 4, "stft_n_fft": 2048, "stft_hop_length": 512, "stft_win_length": 2048, "stft_normalized": False,
         "linear_transformer_depth": 0, "mask_estimator_depth": 2 if model_type == "bs_roformer" else 1}
    if model_type == "bs_roformer":
        a["freqs_per_bands"] = DEFAULT_FREQS_PER_BANDS
    else:
        a.update(num_bands=60, sample_rate=44100)
    for k, v in model_cfg.items():
        if k in a or k in ("dim", "depth"):
            a[k] = tuple(v) if k == "freqs_per_bands" else v
    if model_cfg.get("stft_window_fn") is not None:       # a callable, or the dotted name a YAML carries (mdxc.stft_window_table)
        a["stft_window_fn"] = model_cfg["stft_window_fn"]
    if a["linear_transformer_depth"]:
        raise NotImplementedError("linear_transformer_depth > 0")
    return a


def read_checkpoint(model_path: str) -> dict:
    """torch.load + the 'state_dict' / 'model' unwrapping of roformer_loader.py:97-104 (torch is the weight container)."""
    from .model_files import safe_torch_load
    sd = safe_torch_load(model_path)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    elif isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]
    return sd


class LoadResult:
    """The fields of roformer/model_loading_result.py:ModelLoadingResult that MDXCSeparator.load_model reads."""

    def __init__(self, success, model_type=None, args=None, state_dict=None, config=None, error_message=None):
        self.success = success
        self.model_type = model_type
        self.args = args
        self.state_dict = state_dict
        self.config = config
        self.error_message = error_message
        self.model = None          # the engine owns the network; there is no torch module


class RoformerLoader:
    """Same public surface as roformer_loader.py:RoformerLoader (load_model, validate_configuration, get_loading_stats,
    reset_loading_stats, detect_model_type, get_default_configuration)."""

    def __init__(self):
        self._loading_stats = {"new_implementation_success": 0, "total_failures": 0}

    def load_model(self, model_path: str, config: dict, device: str = "cpu", state_dict: dict | None = None) -> LoadResult:
        try:
            model_type = model_type_from_path(config, model_path)
            cfg = normalize_config(config, model_type, apply_defaults=True, validate=True)
            model_type = detect_model_type(cfg)
        except ParameterValidationError as e:
            return LoadResult(False, error_message=f"Config validation: {e}")
        try:
            # A Mel-Band YAML whose file name does not say so is typed "bs_roformer" by the normaliser (the nested `model`
            # section is invisible to its detection and the BS defaults bring a freqs_per_bands along); the reference then
            # fails to build the BS model and its legacy path constructs the class straight from the raw `model` section
            # (roformer_loader.py:197-236).  That outcome is taken directly.
            raw_model = config.get("model", config) if isinstance(config, dict) else {}
            legacy = isinstance(raw_model, dict) and ("num_bands" in raw_model) != ("freqs_per_bands" in raw_model) and \
                (("num_bands" in raw_model) != (model_type == "mel_band_roformer"))
            if legacy:
                model_type = "mel_band_roformer" if "num_bands" in raw_model else "bs_roformer"
                args = legacy_constructor_args(raw_model, model_type)
            else:
                args = constructor_args(cfg, model_type)
            if state_dict is None and os.path.exists(model_path):
                state_dict = read_checkpoint(model_path)
            self._loading_stats["new_implementation_success"] += 1
            return LoadResult(True, model_type, args, state_dict, cfg)
        except (RuntimeError, ValueError, TypeError, KeyError) as e:
            self._loading_stats["total_failures"] += 1
            return LoadResult(False, error_message=f"New implementation failed: {e}")

    def validate_configuration(self, config: dict, model_type: str) -> bool:
        try:
            normalize_config(config, model_type, apply_defaults=False, validate=True)
            return True
        except (ParameterValidationError, RuntimeError, ValueError):
            return False

    def get_loading_stats(self) -> dict:
        return dict(self._loading_stats)

    def reset_loading_stats(self) -> None:
        self._loading_stats = {"new_implementation_success": 0, "total_failures": 0}

    def detect_model_type(self, model_path: str) -> str:
        """roformer_loader.py:246-256 (path based)."""
        p = model_path.lower()
        if any(s in p for s in ("bs_roformer", "bs-roformer", "bsroformer")):
            return "bs_roformer"
        if any(s in p for s in ("mel_band_roformer", "mel-band-roformer", "melband")):
            return "mel_band_roformer"
        if "roformer" in p:
            return "bs_roformer"
        raise ValueError(f"Cannot determine Roformer model type from path: {model_path}")

    def get_default_configuration(self, model_type: str) -> dict:
        """roformer_loader.py:258-305."""
        base = {"dim": 512, "depth": 12, "stereo": False, "num_stems": 2, "time_transformer_depth": 2, "freq_transformer_depth": 2,
                "dim_head": 64, "heads": 8, "attn_dropout": 0.0, "ff_dropout": 0.0, "flash_attn": True, "mlp_expansion_factor": 4,
                "sage_attention": False, "zero_dc": True, "use_torch_checkpoint": False, "skip_connection": False}
        if model_type == "bs_roformer":
            base.update(freqs_per_bands=(2, 4, 8, 16, 32, 64), mask_estimator_depth=2, stft_n_fft=2048, stft_hop_length=512,
                        stft_win_length=2048)
        elif model_type == "mel_band_roformer":
            base.update(num_bands=64, sample_rate=44100)
        else:
            raise ValueError(f"Unknown model type: {model_type}")
        return base
