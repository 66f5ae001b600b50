"""Host-side mirror of the reference's MDX plugin surface for the demix path.

Same names, argument meaning and error behaviour as the reference
(audio_separator/separator/uvr_lib_v5/stft.py and
audio_separator/separator/architectures/mdx_separator.py), but every numerical
step runs in the HIP engine (libasx.so).  File decoding / stem writing stay with
the reference's CommonSeparator (out of scope).  There is no CPU
path in this module.
"""
from __future__ import annotations

import logging

import numpy as np

from .engine import Engine, MDXConfig, NetConfig
from .weights import fold_convtdf_state


def _to_numpy(x):
    if hasattr(x, "detach"):
        return x.detach().cpu().numpy(), True
    return np.asarray(x), False


def _like(arr, was_torch):
    if was_torch:
        import torch
        return torch.from_numpy(arr)
    return arr


class STFT:
    """Drop-in for uvr_lib_v5/stft.py:STFT (ctor :11, __call__ :20, inverse :99).

    Accepts torch tensors or numpy arrays of shape [..., 2, time]; results come
    back as the same kind of object (host memory).
    """

    def __init__(self, logger, n_fft, hop_length, dim_f, device, engine: Engine | None = None):
        self.logger = logger
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.dim_f = dim_f
        self.device = device
        if engine is None:
            # any chunk geometry works for the stage hooks; use the smallest legal one
            seg = n_fft // hop_length + 2
            engine = Engine(MDXConfig(n_fft=n_fft, hop_length=hop_length, dim_f=dim_f, segment_size=seg, overlap=0.25),
                            device=_device_index(device))
        self.engine = engine

    def __call__(self, input_tensor):
        x, was_torch = _to_numpy(input_tensor)
        batch_dims = x.shape[:-2]
        ch, t = x.shape[-2:]
        if ch != 2:
            raise ValueError(f"Expected a 2-channel audio signal, but got {ch} channels")
        out = self.engine.stft(x.reshape(-1, 2, t))
        return _like(out.reshape(*batch_dims, 4, self.dim_f, out.shape[-1]), was_torch)

    def calculate_inverse_dimensions(self, input_tensor):
        batch_dimensions = tuple(input_tensor.shape[:-3])
        channel_dim, freq_dim, time_dim = input_tensor.shape[-3:]
        return batch_dimensions, channel_dim, freq_dim, time_dim, self.n_fft // 2 + 1

    def inverse(self, input_tensor):
        x, was_torch = _to_numpy(input_tensor)
        batch_dims, channel_dim, freq_dim, time_dim, _ = self.calculate_inverse_dimensions(x)
        if channel_dim != 4 or freq_dim != self.dim_f:
            raise ValueError(f"Expected [..., 4, {self.dim_f}, T], got {x.shape}")
        out = self.engine.istft(x.reshape(-1, 4, freq_dim, time_dim))
        return _like(out.reshape(*batch_dims, 2, -1), was_torch)


def _device_index(device) -> int:
    """GPU ordinal of ``common_config["torch_device"]``: a torch.device, "cuda" / "cuda:N", an int, or None (-> 0)."""
    if isinstance(device, bool) or device is None:
        return 0
    if isinstance(device, int):
        return device
    if isinstance(device, str):
        tail = device.rsplit(":", 1)[1] if ":" in device else ""
        return int(tail) if tail.isdigit() else 0
    idx = getattr(device, "index", None)
    return int(idx) if isinstance(idx, int) else 0


class MDXDemixer:
    """The demix path of MDXSeparator (mdx_separator.py:16) on the HIP engine.

    ``common_config`` / ``arch_config`` carry the keys the reference's
    ``Separator.load_model`` builds (separator.py:867-886, :125).  The weights
    come from the ``.onnx`` file at ``common_config["model_path"]`` (onnx_reader.py)
    or, when given, from a ConvTDFNet ``state_dict`` (torch as the weight container).
    The file-level plugin class on top of this is architectures/mdx_separator.py.
    """

    def __init__(self, common_config: dict, arch_config: dict, state_dict: dict | None = None,
                 net_config: NetConfig | None = None, max_batch: int = 0):
        self.logger = common_config.get("logger") or logging.getLogger(__name__)
        self.torch_device = common_config.get("torch_device")
        self.model_data = common_config.get("model_data") or {}
        self.model_path = common_config.get("model_path")
        self.normalization_threshold = common_config.get("normalization_threshold", 0.9)
        self.amplification_threshold = common_config.get("amplification_threshold", 0.0)
        self.invert_using_spec = common_config.get("invert_using_spec", False)

        self.segment_size = arch_config.get("segment_size")
        self.overlap = arch_config.get("overlap")
        self.batch_size = arch_config.get("batch_size", 1)
        self.hop_length = arch_config.get("hop_length")
        self.enable_denoise = arch_config.get("enable_denoise")

        self.compensate = self.model_data["compensate"]
        self.dim_f = self.model_data["mdx_dim_f_set"]
        self.dim_t = 2 ** self.model_data["mdx_dim_t_set"]
        self.n_fft = self.model_data["mdx_n_fft_scale_set"]

        self.engine = Engine(MDXConfig(n_fft=self.n_fft, hop_length=self.hop_length, dim_f=self.dim_f,
                                       segment_size=self.segment_size, overlap=float(self.overlap),
                                       enable_denoise=bool(self.enable_denoise), max_batch=max_batch),
                             device=_device_index(self.torch_device))
        self.n_bins = 0
        self.trim = 0
        self.chunk_size = 0
        self.gen_size = 0
        self.stft = None
        self.primary_source = None
        self.secondary_source = None
        if state_dict is not None or self.model_path:
            self.load_model(state_dict, net_config)

    def load_model(self, state_dict: dict | None = None, net_config: NetConfig | None = None):
        """Replaces ort.InferenceSession(model_path) (mdx_separator.py:108-133).

        With no ``state_dict`` the ``.onnx`` file at ``common_config["model_path"]`` is read
        (onnx_reader.convtdf_from_onnx); a ConvTDFNet ``state_dict`` (torch as the weight
        container) takes precedence when given."""
        if self.segment_size != self.dim_t:
            # the reference falls back to onnx2torch here; the graph is fully convolutional
            # except for the TDF linears, whose width is tied to dim_f only, so any segment works
            self.logger.warning("segment_size != dim_t: running the net on the requested segment size")
        if state_dict is None:
            from .onnx_reader import convtdf_from_onnx
            net_config, tensors = convtdf_from_onnx(self.model_path, dim_t=self.segment_size)
        else:
            if net_config is None:
                net_config = NetConfig(dim_f=self.dim_f, dim_t=self.segment_size)
            from .weights import state_norm_kind
            if state_norm_kind(state_dict) != net_config.norm:     # GroupNorm(2, c) checkpoints (optimizer 'adamw') carry no running statistics
                from dataclasses import replace
                net_config = replace(net_config, norm=state_norm_kind(state_dict))
            tensors = fold_convtdf_state(state_dict, net_config.num_blocks, net_config.l, net_config.tdf_bias, norm=net_config.norm)
        if net_config.dim_t != self.segment_size or net_config.dim_f != self.dim_f:
            raise ValueError("net dim_t/dim_f must match segment_size/dim_f")
        self.engine.load_net(net_config, tensors)
        self.net_config = net_config

    def initialize_model_settings(self):
        """mdx_separator.py:205-228."""
        self.n_bins = self.n_fft // 2 + 1
        self.trim = self.n_fft // 2
        self.chunk_size = self.hop_length * (self.segment_size - 1)
        self.gen_size = self.chunk_size - 2 * self.trim
        self.stft = STFT(self.logger, self.n_fft, self.hop_length, self.dim_f, self.torch_device, engine=self.engine)

    def demix(self, mix, is_match_mix=False):
        """mdx_separator.py:293-412: float32 [2, N] -> float32 [2, N]."""
        self.initialize_model_settings()
        mix = np.asarray(mix)
        if mix.ndim != 2 or mix.shape[0] != 2:
            raise ValueError(f"Expected a 2-channel audio signal, but got {mix.shape[0] if mix.ndim else 0} channels")
        if mix.shape[1] == 0:
            raise ValueError("Audio file is empty or not valid")
        return self.engine.demix(mix.astype(np.float32, copy=False), is_match_mix=is_match_mix)

    def run_model(self, mix, is_match_mix=False):
        """mdx_separator.py:414-450: [B, 2, chunk_size] -> [B, 2, chunk_size] (numpy)."""
        x, _ = _to_numpy(mix)
        return self.engine.run_model(x.astype(np.float32, copy=False), is_match_mix=is_match_mix)

    def separate_stems(self, mix: np.ndarray):
        """The array part of MDXSeparator.separate (mdx_separator.py:155-182).

        ``mix`` [2, N] is normalised in place like the reference's
        spec_utils.normalize call; returns (primary [N,2], secondary [N,2]).
        """
        self.initialize_model_settings()
        if mix.ndim != 2 or mix.shape[0] != 2:
            raise ValueError(f"Expected a 2-channel audio signal, but got {mix.shape[0] if mix.ndim else 0} channels")
        if mix.shape[1] == 0:
            raise ValueError("Audio file is empty or not valid")
        self.primary_source, self.secondary_source = self.engine.separate(
            mix, self.normalization_threshold, self.amplification_threshold, self.compensate)
        if self.invert_using_spec:
            # The reference hands primary to spec_utils.invert_stem as [N, 2] (mdx_separator.py:177-179), which cannot run
            # (librosa.stft over a 2-sample axis, then a shape mismatch); the intended call -- stem as [2, N] -- runs here.
            raw_mix = self.engine.demix(mix, is_match_mix=True)
            self.secondary_source = self.engine.invert_stem(raw_mix, (self.primary_source * self.compensate).T)
        return self.primary_source, self.secondary_source
