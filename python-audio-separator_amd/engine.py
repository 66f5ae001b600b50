"""ctypes binding of libasx.so (C ABI: include/asx.h).

This is the only bridge between host Python and the HIP engine.  There is no CPU
fallback: if the shared library is missing or no GPU is visible, construction
raises.  PyTorch is not needed here; device pointers are plain integers (e.g.
``tensor.data_ptr()``).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

ASX_FLAG_MATCH_MIX = 1
PROF_CLASSES = ["stft", "conv3x3", "tdf", "down", "up", "conv1x1", "istft", "ola", "finalize", "misc"]


class AsxError(RuntimeError):
    pass


def lib_path() -> str:
    return os.path.join(_HERE, "libasx.so")


class _MdxCfg(C.Structure):
    _fields_ = [("n_fft", C.c_int32), ("hop_length", C.c_int32), ("dim_f", C.c_int32), ("segment_size", C.c_int32),
                ("overlap", C.c_double), ("enable_denoise", C.c_int32), ("max_batch", C.c_int32), ("win_length", C.c_int32),
                ("reserved", C.c_int32)]


class _NetCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim_c", "dim_f", "dim_t", "g", "l", "num_blocks", "k", "bn", "tdf_bias", "norm")]


class _V3Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_channels", "num_subbands", "num_scales", "num_blocks_per_scale",
                                         "num_channels_model", "growth", "bottleneck_factor", "norm", "act",
                                         "num_targets")]


class _RofCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "depth", "heads", "dim_head", "num_stems", "time_depth", "freq_depth",
                                         "mlp_expansion_factor", "mask_estimator_depth", "n_bands", "n_out")] + \
               [("freqs_per_bands", C.c_int32 * 128), ("mel", C.c_int32), ("band_start", C.c_int32 * 128), ("stft_normalized", C.c_int32)]


class _HtCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_sources", "channels", "growth", "nfft", "depth", "kernel_size", "stride",
                                         "dconv_depth", "dconv_comp", "bottom_channels", "t_layers", "t_heads",
                                         "t_hidden", "samplerate", "segment_samples")] + \
               [("freq_emb_scale", C.c_float), ("max_batch", C.c_int32)]


class _HdCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_sources", "channels", "growth", "nfft", "depth", "kernel_size", "stride", "time_stride",
                                         "norm_starts", "norm_groups", "dconv_depth", "dconv_comp", "dconv_attn", "dconv_lstm",
                                         "samplerate", "segment_samples")] + \
               [("freq_emb_scale", C.c_float), ("max_batch", C.c_int32)]


VR_RES_TYPES = {"polyphase": 0, "sinc_fastest": 1}      # ASX_VR_RES_*


class _VrBand(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("sr", "hl", "n_fft", "crop_start", "crop_stop", "hpf_start", "hpf_stop", "lpf_start",
                                         "lpf_stop", "convert", "res_type")]


class _VrCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("bins", "n_bands", "pre_filter_start", "pre_filter_stop", "channel_mode", "arch")] + \
               [("cap", C.c_int32 * 6), ("window_size", C.c_int32), ("offset", C.c_int32), ("max_batch", C.c_int32),
                ("v51", C.c_int32), ("synth_res_type", C.c_int32), ("band", _VrBand * 8)]


class _VrParams(C.Structure):
    _fields_ = [("aggr_value", C.c_float), ("split_bin", C.c_int32), ("is_non_accom", C.c_int32), ("has_corr", C.c_int32),
                ("corr_left", C.c_float), ("corr_right", C.c_float), ("enable_tta", C.c_int32),
                ("enable_post_process", C.c_int32), ("post_thres", C.c_float), ("high_end_process", C.c_int32)]


class _Plan(C.Structure):
    _fields_ = [("n_samples", C.c_int64), ("padded_len", C.c_int64), ("chunk_size", C.c_int64),
                ("gen_size", C.c_int64), ("pad", C.c_int64), ("step", C.c_int64), ("trim", C.c_int32),
                ("n_chunks", C.c_int32), ("n_frames", C.c_int32), ("reserved", C.c_int32)]


class _Profile(C.Structure):
    _fields_ = [("launches", C.c_int64 * 10), ("ms", C.c_double * 10), ("flops", C.c_double * 10),
                ("bytes", C.c_double * 10)]


@dataclass
class MDXConfig:
    """Scalars of MDXSeparator that shape the path (mdx_separator.py:31-72)."""
    n_fft: int = 6144
    hop_length: int = 1024
    dim_f: int = 3072
    segment_size: int = 256
    overlap: float = 0.25
    enable_denoise: bool = False
    max_batch: int = 0
    win_length: int = 0       # torch.stft win_length (Roformer stft_win_length); 0 = n_fft


@dataclass
class NetConfig:
    """ConvTDFNet hyper-parameters (uvr_lib_v5/mdxnet.py:31-44)."""
    dim_c: int = 4
    dim_f: int = 3072
    dim_t: int = 256
    g: int = 48
    l: int = 3
    num_blocks: int = 11
    k: int = 3
    bn: int = 8              # 0 = one Linear(f, f) TDF (modules.py:55-60); -1 = `bn is None`, no TDF branch
    tdf_bias: bool = False
    norm: str = "batch"      # "batch" (optimizer 'rmsprop', BatchNorm folded by weights.fold_convtdf_state) | "group" (GroupNorm(2, c), 'adamw')


@dataclass
class V3Config:
    """TFC_TDF_net hyper-parameters (uvr_lib_v5/tfc_tdf_v3.py:151-214; the model YAML's
    audio/model/training sections).  n_fft / hop_length / dim_f / dim_t live in MDXConfig."""
    num_channels: int = 2
    num_subbands: int = 4
    num_scales: int = 5
    num_blocks_per_scale: int = 2
    num_channels_model: int = 128
    growth: int = 128
    bottleneck_factor: int = 4
    norm: str | None = "InstanceNorm"
    act: str = "gelu"
    num_targets: int = 2


@dataclass
class RofConfig:
    """BSRoformer constructor arguments (roformer_loader.py:123-150); n_fft / hop / dim_t live in MDXConfig."""
    dim: int = 512
    depth: int = 12
    heads: int = 8
    dim_head: int = 64
    num_stems: int = 1
    time_transformer_depth: int = 1
    freq_transformer_depth: int = 1
    mlp_expansion_factor: int = 4
    mask_estimator_depth: int = 2
    freqs_per_bands: tuple = ()
    n_out: int = 2
    mel: bool = False           # MelBandRoformer: band j covers bins [band_starts[j], + freqs_per_bands[j])
    band_starts: tuple = ()
    stft_normalized: bool = False   # torch.stft / istft normalized=True (bs_roformer.py:332, 384)


@dataclass
class HTConfig:
    """HTDemucs constructor arguments (uvr_lib_v5/demucs/htdemucs.py:32-110) the engine builds."""
    sources: tuple = ("drums", "bass", "other", "vocals")
    channels: int = 48
    growth: int = 2
    nfft: int = 4096
    depth: int = 4
    kernel_size: int = 8
    stride: int = 4
    dconv_depth: int = 2
    dconv_comp: int = 8
    freq_emb: float = 0.2
    bottom_channels: int = 0
    t_layers: int = 5
    t_heads: int = 8
    t_hidden_scale: float = 4.0
    samplerate: int = 44100
    segment: object = 7.8          # seconds (Fraction in the checkpoints)
    max_batch: int = 0

    @property
    def segment_samples(self) -> int:
        return int(self.segment * self.samplerate)

    @property
    def transformer_dim(self) -> int:
        return self.bottom_channels or self.channels * self.growth ** (self.depth - 1)


@dataclass
class HDConfig:
    """HDemucs constructor arguments (uvr_lib_v5/demucs/hdemucs.py:362-420) the engine builds (Demucs v3: `hdemucs_mmi`)."""
    sources: tuple = ("drums", "bass", "other", "vocals")
    channels: int = 48
    growth: int = 2
    nfft: int = 4096
    depth: int = 6
    kernel_size: int = 8
    stride: int = 4
    time_stride: int = 2
    norm_starts: int = 4
    norm_groups: int = 4
    dconv_depth: int = 2
    dconv_comp: int = 4
    dconv_attn: int = 4
    dconv_lstm: int = 4
    freq_emb: float = 0.2
    samplerate: int = 44100
    segment: object = 40           # seconds
    max_batch: int = 0

    @property
    def segment_samples(self) -> int:
        return int(self.samplerate * self.segment)


_FP = C.POINTER(C.c_float)
_lib = None

ABI_VERSION = 7   # ASX_ABI_VERSION of include/asx.h the structures below mirror

# every symbol include/asx.h declares
SYMBOLS = ["asx_abi_version", "asx_last_error", "asx_device_count", "asx_engine_create", "asx_engine_destroy",
           "asx_net_begin", "asx_net_set_tensor", "asx_net_commit", "asx_net_flops", "asx_plan_query", "asx_demix",
           "asx_demix_dev", "asx_demix_chunks_dev", "asx_finalize_dev", "asx_separate", "asx_separate_dev",
           "asx_stft", "asx_istft", "asx_net_forward", "asx_run_model", "asx_op_conv", "asx_op_tdf",
           "asx_profile_enable", "asx_profile_read", "asx_v3_begin", "asx_v3_commit", "asx_v3_flops",
           "asx_v3_forward", "asx_mdxc_plan", "asx_mdxc_demix", "asx_mdxc_demix_dev", "asx_set_option",
           "asx_rof_begin", "asx_rof_commit", "asx_rof_flops", "asx_rof_forward", "asx_rof_demix",
           "asx_rof_demix_dev", "asx_ht_begin", "asx_ht_commit", "asx_ht_flops", "asx_ht_forward", "asx_ht_demix",
           "asx_ht_demix_dev", "asx_vr_begin", "asx_vr_commit", "asx_vr_flops", "asx_vr_plan", "asx_vr_forward",
           "asx_vr_analysis", "asx_vr_separate", "asx_vr_separate_dev", "asx_debug_fetch", "asx_mdxc_chunks_dev",
           "asx_mdxc_finalize_dev", "asx_rof_plan", "asx_rof_chunks_dev", "asx_rof_finalize_dev", "asx_ht_plan",
           "asx_ht_segments_dev", "asx_ht_fold_dev", "asx_hd_begin", "asx_hd_commit", "asx_hd_flops",
           "asx_hd_forward", "asx_hd_demix", "asx_hd_demix_dev", "asx_hd_plan", "asx_hd_segments_dev",
           "asx_hd_fold_dev", "asx_pcm16", "asx_pcm16_dev", "asx_pcm16_rows_dev", "asx_pcm_decode_dev",
           "asx_ht_standardize_dev", "asx_ht_bag_accumulate_dev", "asx_ht_bag_finish_dev", "asx_ensemble",
           "asx_ensemble_dev", "asx_invert_stem", "asx_normalize", "asx_normalize_dev", "asx_residual_dev",
           "asx_profile_launches", "asx_debug_trace", "asx_resample_sinc", "asx_resample_sinc_dev", "asx_counter",
           "asx_set_stft_window", "asx_op_tdf_block"]


class _LaunchRec(C.Structure):     # struct asx_launch_rec
    _fields_ = [("cls", C.c_int32), ("ms", C.c_float), ("flops", C.c_double), ("bytes", C.c_double)]


def load_library():
    """dlopen libasx.so and declare the signatures.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    try:
        # torch-ROCm bundles its own libamdhip64 (same soname).  Importing it first makes the
        # dynamic loader resolve libasx.so against that one runtime, so device pointers taken
        # from torch tensors and the engine's own allocations live in the same HIP context.
        import torch  # noqa: F401
    except Exception:  # torch is plumbing, not a requirement of the engine
        pass
    if not os.path.exists(path):
        raise AsxError(f"{path} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                       "There is no CPU fallback for the demix path.")
    lib = C.CDLL(path)
    vp, i32, i64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
    lib.asx_abi_version.restype = C.c_int
    if lib.asx_abi_version() != ABI_VERSION:
        raise AsxError(f"{path} speaks ABI {lib.asx_abi_version()}, this binding ABI {ABI_VERSION}: rebuild (python __graft_entry__.py)")
    lib.asx_last_error.restype = C.c_char_p
    lib.asx_device_count.restype = C.c_int
    lib.asx_engine_create.argtypes = [C.c_int, C.POINTER(_MdxCfg), C.POINTER(vp)]
    lib.asx_engine_destroy.argtypes = [vp]
    lib.asx_engine_destroy.restype = None
    lib.asx_net_begin.argtypes = [vp, C.POINTER(_NetCfg)]
    lib.asx_net_set_tensor.argtypes = [vp, C.c_char_p, _FP, i64]
    lib.asx_net_commit.argtypes = [vp]
    lib.asx_net_flops.argtypes = [vp, i32]
    lib.asx_net_flops.restype = C.c_double
    lib.asx_plan_query.argtypes = [vp, i64, u32, C.POINTER(_Plan)]
    lib.asx_demix.argtypes = [vp, _FP, i64, _FP, u32]
    lib.asx_demix_dev.argtypes = [vp, vp, i64, vp, u32, vp]
    lib.asx_demix_chunks_dev.argtypes = [vp, vp, i64, i32, i32, vp, u32, vp]
    lib.asx_finalize_dev.argtypes = [vp, vp, i64, vp, u32, vp]
    f32 = C.c_float
    lib.asx_separate.argtypes = [vp, _FP, i64, f32, f32, i32, f32, _FP, _FP]
    lib.asx_separate_dev.argtypes = [vp, vp, i64, f32, f32, i32, f32, vp, vp, vp]
    lib.asx_stft.argtypes = [vp, _FP, i32, i64, _FP]
    lib.asx_istft.argtypes = [vp, _FP, i32, i32, _FP]
    lib.asx_net_forward.argtypes = [vp, _FP, i32, _FP]
    lib.asx_run_model.argtypes = [vp, _FP, i32, _FP, u32]
    lib.asx_op_conv.argtypes = [vp, C.c_char_p, _FP, i32, i32, i32, i32, _FP, _FP, i32, _FP, i32, _FP]
    lib.asx_op_tdf.argtypes = [vp, _FP, i32, i32, i32, i32, _FP, _FP, i32, _FP, _FP, _FP, _FP]
    lib.asx_op_tdf_block.argtypes = [vp, _FP, i32, i32, i32, i32, _FP, _FP, _FP, i32, _FP, _FP, _FP, _FP, _FP]
    lib.asx_v3_begin.argtypes = [vp, C.POINTER(_V3Cfg)]
    lib.asx_v3_commit.argtypes = [vp]
    lib.asx_v3_flops.argtypes = [vp, i32]
    lib.asx_v3_flops.restype = C.c_double
    lib.asx_v3_forward.argtypes = [vp, _FP, i32, _FP]
    lib.asx_mdxc_plan.argtypes = [vp, i64, i32, C.POINTER(_Plan)]
    lib.asx_mdxc_demix.argtypes = [vp, _FP, i64, i32, _FP]
    lib.asx_mdxc_demix_dev.argtypes = [vp, vp, i64, i32, vp, vp]
    lib.asx_rof_begin.argtypes = [vp, C.POINTER(_RofCfg)]
    lib.asx_rof_commit.argtypes = [vp]
    lib.asx_rof_flops.argtypes = [vp, i32]
    lib.asx_rof_flops.restype = C.c_double
    lib.asx_rof_forward.argtypes = [vp, _FP, i32, _FP]
    lib.asx_rof_demix.argtypes = [vp, _FP, i64, i64, _FP]
    lib.asx_rof_demix_dev.argtypes = [vp, vp, i64, i64, vp, vp]
    lib.asx_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.asx_counter.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    lib.asx_set_stft_window.argtypes = [vp, _FP, i32]
    lib.asx_ht_begin.argtypes = [vp, C.POINTER(_HtCfg)]
    lib.asx_ht_commit.argtypes = [vp]
    lib.asx_ht_flops.argtypes = [vp]
    lib.asx_ht_flops.restype = C.c_double
    lib.asx_ht_forward.argtypes = [vp, _FP, i32, i64, _FP]
    lib.asx_ht_demix.argtypes = [vp, _FP, i64, i32, C.POINTER(C.c_int64), C.c_double, u32, _FP]
    lib.asx_ht_demix_dev.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_int64), C.c_double, u32, vp, vp]
    lib.asx_vr_begin.argtypes = [vp, C.POINTER(_VrCfg)]
    lib.asx_vr_commit.argtypes = [vp]
    lib.asx_vr_flops.argtypes = [vp]
    lib.asx_vr_flops.restype = C.c_double
    lib.asx_vr_plan.argtypes = [vp, i64, C.POINTER(i32), C.POINTER(i64)]
    lib.asx_vr_forward.argtypes = [vp, _FP, i32, _FP]
    lib.asx_vr_analysis.argtypes = [vp, _FP, i64, _FP]
    lib.asx_vr_separate.argtypes = [vp, _FP, i64, C.POINTER(_VrParams), _FP, _FP]
    lib.asx_vr_separate_dev.argtypes = [vp, vp, i64, C.POINTER(_VrParams), vp, vp, vp]
    lib.asx_debug_fetch.argtypes = [vp, C.c_char_p, _FP, i64]
    lib.asx_resample_sinc.argtypes = [vp, _FP, i32, i64, C.c_double, i32, _FP, i64]
    lib.asx_resample_sinc_dev.argtypes = [vp, vp, i32, i64, C.c_double, i32, vp, i64, vp]
    lib.asx_ensemble.argtypes = [vp, _FP, i32, i64, i32, C.POINTER(C.c_double), _FP, C.POINTER(i64)]
    lib.asx_ensemble_dev.argtypes = [vp, vp, i32, i64, i32, C.POINTER(C.c_double), vp, C.POINTER(i64), vp]
    lib.asx_invert_stem.argtypes = [vp, _FP, _FP, i64, _FP, C.POINTER(i64)]
    lib.asx_pcm16.argtypes = [vp, _FP, i64, C.c_float, C.c_float, i32, C.POINTER(C.c_int16), _FP]
    lib.asx_normalize.argtypes = [vp, _FP, i64, C.c_float, C.c_float, i32, _FP]
    lib.asx_pcm16_dev.argtypes = [vp, vp, i64, C.c_float, C.c_float, i32, vp, _FP, vp]
    lib.asx_pcm16_rows_dev.argtypes = [vp, vp, i64, C.c_float, C.c_float, i32, vp, _FP, vp]
    lib.asx_pcm_decode_dev.argtypes = [vp, vp, i64, i32, i32, vp, _FP, vp]
    lib.asx_profile_launches.argtypes = [vp, C.POINTER(_LaunchRec), i32, C.POINTER(i32)]
    lib.asx_normalize_dev.argtypes = [vp, vp, i64, C.c_float, C.c_float, i32, vp]
    lib.asx_residual_dev.argtypes = [vp, vp, vp, vp, i64, vp]
    lib.asx_ht_standardize_dev.argtypes = [vp, vp, i64, vp, vp]
    lib.asx_ht_bag_accumulate_dev.argtypes = [vp, vp, vp, _FP, i32, i64, i32, vp]
    lib.asx_ht_bag_finish_dev.argtypes = [vp, vp, _FP, i32, vp, i64, C.c_uint32, vp, vp]
    lib.asx_mdxc_chunks_dev.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp]
    lib.asx_mdxc_finalize_dev.argtypes = [vp, vp, i64, i32, vp, vp]
    lib.asx_rof_plan.argtypes = [vp, i64, i64, C.POINTER(i32), C.POINTER(i64)]
    lib.asx_rof_chunks_dev.argtypes = [vp, vp, i64, i64, i32, i32, vp, vp]
    lib.asx_rof_finalize_dev.argtypes = [vp, vp, i64, i64, vp, vp]
    lib.asx_ht_plan.argtypes = [vp, i64, i32, C.POINTER(C.c_int64), C.c_double, C.POINTER(i32), C.POINTER(i64)]
    lib.asx_ht_segments_dev.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_int64), C.c_double, u32, i32, i32, vp, vp]
    lib.asx_ht_fold_dev.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_int64), C.c_double, u32, vp, vp, vp]
    lib.asx_hd_begin.argtypes = [vp, C.POINTER(_HdCfg)]
    lib.asx_hd_commit.argtypes = [vp]
    lib.asx_hd_flops.argtypes = [vp, i64]
    lib.asx_hd_flops.restype = C.c_double
    lib.asx_hd_forward.argtypes = [vp, _FP, i32, i64, _FP]
    lib.asx_hd_demix.argtypes = [vp, _FP, i64, i32, C.POINTER(C.c_int64), C.c_double, u32, _FP]
    lib.asx_hd_demix_dev.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_int64), C.c_double, u32, vp, vp]
    lib.asx_hd_plan.argtypes = [vp, i64, i32, C.POINTER(C.c_int64), C.c_double, C.POINTER(i32), C.POINTER(i64)]
    lib.asx_hd_segments_dev.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_int64), C.c_double, u32, i32, i32, vp, vp]
    lib.asx_hd_fold_dev.argtypes = [vp, vp, i64, i32, C.POINTER(C.c_int64), C.c_double, u32, vp, vp, vp]
    lib.asx_profile_enable.argtypes = [vp, i32]
    lib.asx_profile_read.argtypes = [vp, C.POINTER(_Profile)]
    for name in SYMBOLS:
        getattr(lib, name)  # AttributeError if the library does not export what the header declares
    _lib = lib
    return lib


def ht_pos_tables(hc: "HTConfig") -> dict:
    """The two sinusoidal tables of CrossTransformerEncoder (transformer.py:18-46, 522-537) in torch float32 on the
    host, in the engine's token order: "pos_emb_freq" rows (t1, fr), "pos_emb_time" rows t2."""
    import math

    import torch
    Ct = hc.transformer_dim
    hop = hc.nfft // 4
    T1 = -(-hc.segment_samples // hop)
    Fr = hc.nfft // 2 // hc.stride ** hc.depth
    T2 = hc.segment_samples
    for _ in range(hc.depth):
        T2 = -(-T2 // hc.stride)
    pe = torch.zeros(Ct, Fr, T1)
    dm = Ct // 2
    div = torch.exp(torch.arange(0.0, dm, 2) * -(math.log(10000.0) / dm))
    pw = torch.arange(0.0, T1).unsqueeze(1)
    ph = torch.arange(0.0, Fr).unsqueeze(1)
    pe[0:dm:2] = torch.sin(pw * div).transpose(0, 1).unsqueeze(1).repeat(1, Fr, 1)
    pe[1:dm:2] = torch.cos(pw * div).transpose(0, 1).unsqueeze(1).repeat(1, Fr, 1)
    pe[dm::2] = torch.sin(ph * div).transpose(0, 1).unsqueeze(2).repeat(1, 1, T1)
    pe[dm + 1::2] = torch.cos(ph * div).transpose(0, 1).unsqueeze(2).repeat(1, 1, T1)
    half = Ct // 2
    pos = torch.arange(T2).view(-1, 1)
    adim = torch.arange(half).view(1, -1)
    phase = pos / (10000.0 ** (adim / (half - 1)))
    pt = torch.cat([torch.cos(phase), torch.sin(phase)], dim=-1)
    return {"pos_emb_freq": pe.permute(2, 1, 0).reshape(T1 * Fr, Ct).contiguous().numpy(),
            "pos_emb_time": pt.contiguous().numpy()}


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_FP)


def _optptr(a):
    return _ptr(a) if a is not None else None


class Engine:
    """One HIP engine bound to one GPU (asx_engine)."""

    def __init__(self, cfg: MDXConfig, device: int = 0):
        self._lib = load_library()
        self._h = C.c_void_p()
        self._options = {}
        self.cfg = cfg
        self.device = device
        self.net_cfg = None
        if self._lib.asx_device_count() <= 0:
            raise AsxError("no HIP device visible: the demix path runs on MI355X only (no CPU fallback)")
        c = _MdxCfg(cfg.n_fft, cfg.hop_length, cfg.dim_f, cfg.segment_size, float(cfg.overlap),
                    int(bool(cfg.enable_denoise)), int(cfg.max_batch), int(getattr(cfg, "win_length", 0) or 0), 0)
        self._check(self._lib.asx_engine_create(device, C.byref(c), C.byref(self._h)))

    # -- plumbing ---------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            msg = self._lib.asx_last_error()
            raise AsxError(f"asx error {rc}: {msg.decode() if msg else '?'}")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.asx_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key: str, value: int):
        self._check(self._lib.asx_set_option(self._h, key.encode(), int(value)))
        self._options[key] = (1 if int(value) > 0 else 0) if key in ("gemm_bf16x6", "gemm_f16x3", "gemm_pair_images", "conv_down_bf16x6", "conv_up_bf16x6") else max(0, int(value)) if key == "conv_direct_f16x3" else int(value)

    def option(self, key: str) -> int:
        """Current value of an engine option (the library default when it was never set here)."""
        defaults = {"winograd": max(0, int(os.environ.get("ASX_WINOGRAD", "3"))),
                    "winograd_stationary": max(0, int(os.environ.get("ASX_WINOS", "0"))),
                    "gemm_bf16x6": 1 if int(os.environ.get("ASX_GEMM_BF16X6", "1")) > 0 else 0,
                    "gemm_f16x3": 1 if int(os.environ.get("ASX_GEMM_F16X3", "1")) > 0 else 0,
                    "winograd_bf16x6": max(0, int(os.environ.get("ASX_WINO6", "144"))),
                    "conv_direct_f16x3": max(0, int(os.environ.get("ASX_CONV3H", "144"))),
                    "gemm_pair_images": 0,
                    "conv_down_bf16x6": 1 if int(os.environ.get("ASX_DOWN6", "1")) > 0 else 0,
                    "conv_up_bf16x6": 1 if int(os.environ.get("ASX_UP6", "1")) > 0 else 0}
        if key not in defaults:
            raise AsxError(f"unknown engine option {key!r} (known: {sorted(defaults)})")
        return self._options.get(key, defaults[key])

    def counter(self, name: str) -> int:
        """A library launch counter (asx_counter): "tdf3_launches" = bf16x6 row-GEMM launches of this process, ..."""
        out = C.c_int64(0)
        self._check(self._lib.asx_counter(self._h, name.encode(), C.byref(out)))
        return int(out.value)

    # -- weights ------------------------------------------------------------
    def load_net(self, net_cfg: NetConfig, tensors: dict):
        """tensors: canonical name -> float32 array (see include/asx.h)."""
        if net_cfg.norm not in ("batch", "group"):
            raise ValueError(f"NetConfig.norm must be 'batch' or 'group', got {net_cfg.norm!r}")
        n = _NetCfg(net_cfg.dim_c, net_cfg.dim_f, net_cfg.dim_t, net_cfg.g, net_cfg.l, net_cfg.num_blocks, net_cfg.k,
                    -1 if net_cfg.bn is None else net_cfg.bn, int(bool(net_cfg.tdf_bias)), 1 if net_cfg.norm == "group" else 0)
        self._check(self._lib.asx_net_begin(self._h, C.byref(n)))
        for name, arr in tensors.items():
            a = _f32(arr).reshape(-1)
            self._check(self._lib.asx_net_set_tensor(self._h, name.encode(), _ptr(a), a.size))
        self._check(self._lib.asx_net_commit(self._h))
        self.net_cfg = net_cfg

    def net_flops(self, batch: int = 1) -> float:
        return float(self._lib.asx_net_flops(self._h, batch))

    # -- MDXC / TFC-TDF v3 ------------------------------------------------------
    def load_v3(self, v3: V3Config, state_dict: dict):
        """state_dict: the reference TFC_TDF_net's own keys -> float32 arrays / torch tensors."""
        norm = {None: 0, "": 0, "None": 0, "InstanceNorm": 1}.get(v3.norm)
        act = {"relu": 0, "gelu": 1}.get(v3.act)
        if norm is None or act is None:
            raise ValueError(f"unsupported norm/act for the HIP path: {v3.norm!r}/{v3.act!r}")
        c = _V3Cfg(v3.num_channels, v3.num_subbands, v3.num_scales, v3.num_blocks_per_scale, v3.num_channels_model,
                   v3.growth, v3.bottleneck_factor, norm, act, v3.num_targets)
        self._check(self._lib.asx_v3_begin(self._h, C.byref(c)))
        for name, t in state_dict.items():
            if hasattr(t, "detach"):
                t = t.detach().cpu().numpy()
            a = _f32(t).reshape(-1)
            self._check(self._lib.asx_net_set_tensor(self._h, name.encode(), _ptr(a), a.size))
        self._check(self._lib.asx_v3_commit(self._h))
        self.v3_cfg = v3

    def v3_flops(self, batch: int = 1) -> float:
        return float(self._lib.asx_v3_flops(self._h, batch))

    def v3_forward(self, wave: np.ndarray) -> np.ndarray:
        wave = _f32(wave)
        B, ch, Cn = wave.shape
        out = np.empty((B, self.v3_cfg.num_targets, 2, Cn), np.float32)
        self._check(self._lib.asx_v3_forward(self._h, _ptr(wave), B, _ptr(out)))
        return out

    def mdxc_plan(self, n_samples: int, overlap: int) -> dict:
        p = _Plan()
        self._check(self._lib.asx_mdxc_plan(self._h, n_samples, overlap, C.byref(p)))
        return {k: getattr(p, k) for k, _ in _Plan._fields_ if k != "reserved"}

    def mdxc_demix(self, mix: np.ndarray, overlap: int) -> np.ndarray:
        mix = _f32(mix)
        if mix.ndim != 2 or mix.shape[0] != 2:
            raise ValueError(f"Expected a 2-channel audio signal, but got shape {mix.shape}")
        out = np.empty((self.v3_cfg.num_targets, 2, mix.shape[1]), np.float32)
        self._check(self._lib.asx_mdxc_demix(self._h, _ptr(mix), mix.shape[1], int(overlap), _ptr(out)))
        return out

    def mdxc_demix_dev(self, mix_ptr: int, n_samples: int, overlap: int, out_ptr: int, stream: int = 0):
        self._check(self._lib.asx_mdxc_demix_dev(self._h, mix_ptr, n_samples, int(overlap), out_ptr, stream or None))

    # -- BS-Roformer ------------------------------------------------------------
    def load_rof(self, rc: RofConfig, state_dict: dict):
        if len(rc.freqs_per_bands) > 128:
            raise ValueError("at most 128 bands")
        c = _RofCfg(rc.dim, rc.depth, rc.heads, rc.dim_head, rc.num_stems, rc.time_transformer_depth,
                    rc.freq_transformer_depth, rc.mlp_expansion_factor, rc.mask_estimator_depth,
                    len(rc.freqs_per_bands), rc.n_out)
        for i, f in enumerate(rc.freqs_per_bands):
            c.freqs_per_bands[i] = int(f)
        c.mel = int(bool(rc.mel))
        c.stft_normalized = int(bool(rc.stft_normalized))
        if rc.mel:
            if len(rc.band_starts) != len(rc.freqs_per_bands):
                raise ValueError("mel: one band start per band")
            for i, f in enumerate(rc.band_starts):
                c.band_start[i] = int(f)
        self._check(self._lib.asx_rof_begin(self._h, C.byref(c)))
        for name, t in state_dict.items():
            if hasattr(t, "detach"):
                t = t.detach().cpu().numpy()
            a = _f32(t).reshape(-1)
            self._check(self._lib.asx_net_set_tensor(self._h, name.encode(), _ptr(a), a.size))
        self._check(self._lib.asx_rof_commit(self._h))
        self.rof_cfg = rc

    def set_stft_window(self, window: np.ndarray):
        """Replace the periodic Hann of torch.stft / istft by `window` [n_fft] (already zero padded from win_length like torch pads it)."""
        w = _f32(window).reshape(-1)
        self._check(self._lib.asx_set_stft_window(self._h, _ptr(w), int(w.size)))

    def rof_flops(self, batch: int = 1) -> float:
        return float(self._lib.asx_rof_flops(self._h, batch))

    def rof_forward(self, wave: np.ndarray) -> np.ndarray:
        wave = _f32(wave)
        B, ch, Cn = wave.shape
        out = np.empty((B, self.rof_cfg.num_stems, 2, Cn), np.float32)
        self._check(self._lib.asx_rof_forward(self._h, _ptr(wave), B, _ptr(out)))
        return out

    def rof_demix(self, mix: np.ndarray, step: int) -> np.ndarray:
        mix = _f32(mix)
        if mix.ndim != 2 or mix.shape[0] != 2:
            raise ValueError(f"Expected a 2-channel audio signal, but got shape {mix.shape}")
        out = np.empty((self.rof_cfg.n_out, 2, mix.shape[1]), np.float32)
        self._check(self._lib.asx_rof_demix(self._h, _ptr(mix), mix.shape[1], int(step), _ptr(out)))
        return out

    def rof_demix_dev(self, mix_ptr: int, n_samples: int, step: int, out_ptr: int, stream: int = 0):
        self._check(self._lib.asx_rof_demix_dev(self._h, mix_ptr, n_samples, int(step), out_ptr, stream or None))

    # -- Demucs v4 ----------------------------------------------------------------
    def load_ht(self, hc: HTConfig, state_dict: dict, pos_tables: bool = True):
        """HTDemucs(**kwargs) + load_state_dict.  pos_tables: compute the sinusoidal tables with torch on the host
        (bit-identical to transformer.py:18-46 on CPU) and hand them over; otherwise the engine builds them itself."""
        C_t = hc.transformer_dim
        c = _HtCfg(len(hc.sources), hc.channels, hc.growth, hc.nfft, hc.depth, hc.kernel_size, hc.stride, hc.dconv_depth,
                   hc.dconv_comp, hc.bottom_channels, hc.t_layers, hc.t_heads, int(C_t * hc.t_hidden_scale),
                   hc.samplerate, hc.segment_samples, float(hc.freq_emb), hc.max_batch)
        self._check(self._lib.asx_ht_begin(self._h, C.byref(c)))
        tensors = dict(state_dict)
        if pos_tables and hc.t_layers > 0:
            tensors.update(ht_pos_tables(hc))
        for name, t in tensors.items():
            if hasattr(t, "detach"):
                t = t.detach().cpu().numpy()
            a = _f32(t).reshape(-1)
            self._check(self._lib.asx_net_set_tensor(self._h, name.encode(), _ptr(a), a.size))
        self._check(self._lib.asx_ht_commit(self._h))
        self.ht_cfg = hc

    def ht_flops(self) -> float:
        return float(self._lib.asx_ht_flops(self._h))

    def ht_forward(self, mix: np.ndarray) -> np.ndarray:
        mix = _f32(mix)
        B, ch, L = mix.shape
        if ch != 2:
            raise ValueError(f"Expected a 2-channel audio signal, but got {ch} channels")
        out = np.empty((B, len(self.ht_cfg.sources), 2, L), np.float32)
        self._check(self._lib.asx_ht_forward(self._h, _ptr(mix), B, L, _ptr(out)))
        return out

    def ht_demix(self, mix: np.ndarray, shifts: int = 0, offsets=None, overlap: float = 0.25, standardize: bool = False,
                 swap01: bool = False) -> np.ndarray:
        mix = _f32(mix)
        if mix.ndim != 2 or mix.shape[0] != 2:
            raise ValueError(f"Expected a 2-channel audio signal, but got shape {mix.shape}")
        out = np.empty((len(self.ht_cfg.sources), 2, mix.shape[1]), np.float32)
        offs = None
        if shifts:
            if offsets is None or len(offsets) != shifts:
                raise ValueError("shifts > 0 needs one offset per shift")
            offs = (C.c_int64 * shifts)(*[int(o) for o in offsets])
        flags = (1 if standardize else 0) | (2 if swap01 else 0)
        self._check(self._lib.asx_ht_demix(self._h, _ptr(mix), mix.shape[1], int(shifts), offs, float(overlap), flags,
                                           _ptr(out)))
        return out

    def ht_demix_dev(self, mix_ptr: int, n_samples: int, out_ptr: int, shifts: int = 0, offsets=None,
                     overlap: float = 0.25, flags: int = 0, stream: int = 0):
        offs = (C.c_int64 * shifts)(*[int(o) for o in offsets]) if shifts else None
        self._check(self._lib.asx_ht_demix_dev(self._h, mix_ptr, n_samples, int(shifts), offs, float(overlap), flags,
                                               out_ptr, stream or None))

    # -- BagOfModels combine on the device -----------------------------------------
    def ht_standardize_dev(self, mix_ptr: int, n_samples: int, out_ptr: int, stream: int = 0):
        self._check(self._lib.asx_ht_standardize_dev(self._h, mix_ptr, n_samples, out_ptr, stream or None))

    def ht_bag_accumulate_dev(self, est_ptr: int, member_ptr: int, weights, n_samples: int, first: bool, stream: int = 0):
        w = (C.c_float * len(weights))(*[float(v) for v in weights])
        self._check(self._lib.asx_ht_bag_accumulate_dev(self._h, est_ptr, member_ptr, w, len(weights), n_samples, int(bool(first)),
                                                        stream or None))

    def ht_bag_finish_dev(self, est_ptr: int, totals, mix_ptr: int, n_samples: int, out_ptr: int, standardize: bool = True,
                          swap01: bool = True, stream: int = 0):
        t = (C.c_float * len(totals))(*[float(v) for v in totals])
        flags = (1 if standardize else 0) | (2 if swap01 else 0)
        self._check(self._lib.asx_ht_bag_finish_dev(self._h, est_ptr, t, len(totals), mix_ptr, n_samples, flags, out_ptr, stream or None))

    # -- Demucs v3 ----------------------------------------------------------------
    def load_hd(self, hc: HDConfig, state_dict: dict):
        """HDemucs(**kwargs) + load_state_dict (uvr_lib_v5/demucs/hdemucs.py:362-571)."""
        c = _HdCfg(len(hc.sources), hc.channels, hc.growth, hc.nfft, hc.depth, hc.kernel_size, hc.stride, hc.time_stride,
                   hc.norm_starts, hc.norm_groups, hc.dconv_depth, hc.dconv_comp, hc.dconv_attn, hc.dconv_lstm, hc.samplerate,
                   hc.segment_samples, float(hc.freq_emb), hc.max_batch)
        self._check(self._lib.asx_hd_begin(self._h, C.byref(c)))
        for name, t in state_dict.items():
            if hasattr(t, "detach"):
                t = t.detach().cpu().numpy()
            a = _f32(t).reshape(-1)
            self._check(self._lib.asx_net_set_tensor(self._h, name.encode(), _ptr(a), a.size))
        self._check(self._lib.asx_hd_commit(self._h))
        self.hd_cfg = hc

    def hd_flops(self, length: int) -> float:
        return float(self._lib.asx_hd_flops(self._h, int(length)))

    def hd_forward(self, mix: np.ndarray) -> np.ndarray:
        """HDemucs.forward: [B, 2, L] -> [B, S, 2, L], any L >= nfft."""
        mix = _f32(mix)
        B, ch, L = mix.shape
        if ch != 2:
            raise ValueError(f"Expected a 2-channel audio signal, but got {ch} channels")
        out = np.empty((B, len(self.hd_cfg.sources), 2, L), np.float32)
        self._check(self._lib.asx_hd_forward(self._h, _ptr(mix), B, L, _ptr(out)))
        return out

    def hd_demix(self, mix: np.ndarray, shifts: int = 0, offsets=None, overlap: float = 0.25, standardize: bool = False,
                 swap01: bool = False) -> np.ndarray:
        mix = _f32(mix)
        if mix.ndim != 2 or mix.shape[0] != 2:
            raise ValueError(f"Expected a 2-channel audio signal, but got shape {mix.shape}")
        out = np.empty((len(self.hd_cfg.sources), 2, mix.shape[1]), np.float32)
        if shifts and (offsets is None or len(offsets) != shifts):
            raise ValueError("shifts > 0 needs one offset per shift")
        flags = (1 if standardize else 0) | (2 if swap01 else 0)
        self._check(self._lib.asx_hd_demix(self._h, _ptr(mix), mix.shape[1], int(shifts), self._offs(shifts, offsets), float(overlap),
                                           flags, _ptr(out)))
        return out

    def hd_demix_dev(self, mix_ptr: int, n_samples: int, out_ptr: int, shifts: int = 0, offsets=None, overlap: float = 0.25,
                     flags: int = 0, stream: int = 0):
        self._check(self._lib.asx_hd_demix_dev(self._h, mix_ptr, n_samples, int(shifts), self._offs(shifts, offsets), float(overlap),
                                               flags, out_ptr, stream or None))

    def hd_plan(self, n, shifts=0, offsets=None, overlap=0.25):
        k, c = C.c_int32(), C.c_int64()
        self._check(self._lib.asx_hd_plan(self._h, n, int(shifts), self._offs(shifts, offsets), float(overlap), C.byref(k), C.byref(c)))
        return {"n_chunks": k.value, "chunk_size": c.value}

    def hd_segments_dev(self, mix_ptr, n, k0, k1, out_ptr, shifts=0, offsets=None, overlap=0.25, flags=0, stream=0):
        self._check(self._lib.asx_hd_segments_dev(self._h, mix_ptr, n, int(shifts), self._offs(shifts, offsets), float(overlap), flags,
                                                  k0, k1, out_ptr, stream or None))

    def hd_fold_dev(self, mix_ptr, n, chunks_ptr, out_ptr, shifts=0, offsets=None, overlap=0.25, flags=0, stream=0):
        self._check(self._lib.asx_hd_fold_dev(self._h, mix_ptr, n, int(shifts), self._offs(shifts, offsets), float(overlap), flags,
                                              chunks_ptr, out_ptr, stream or None))

    # -- VR -----------------------------------------------------------------------
    def load_vr(self, model_params: dict, arch: int, capacity, state_dict: dict, window_size: int = 512, offset: int = 128,
                max_batch: int = 0, v51=None, wav_resolution: str = "polyphase"):
        """nets.determine_model_capacity(bins * 2, arch) + load_state_dict.  model_params: the modelparams JSON dict
        (int band keys); capacity: the model_capacity_data table of nets.py:74-86.  v51 = (nout, nout_lstm) selects
        nets_new.CascadedNet and the is_v51_model branches instead (capacity is then ignored).  ``wav_resolution``: the
        synthesis chain's converter (spec_utils.py:33-38), "sinc_fastest" or "polyphase"; the analysis converters follow each
        band's own "res_type" ("sinc_fastest" -> the sinc converter, anything else -> polyphase)."""
        mp = model_params
        mode = 3 if mp.get("reverse") else (1 if mp.get("mid_side") else (2 if mp.get("mid_side_b2") else 0))
        nb = len(mp["band"])
        c = _VrCfg(mp["bins"], nb, mp["pre_filter_start"], mp["pre_filter_stop"], 0 if v51 else mode, int(arch))
        caps = (v51[0], v51[1], 0, 0, 0, 0) if v51 else (capacity[0][1], capacity[2][1], capacity[3][1], capacity[4][1], capacity[5][1], 0)
        for i, v in enumerate(caps):
            c.cap[i] = int(v)
        c.window_size, c.offset, c.max_batch, c.v51 = int(window_size), int(offset), int(max_batch), int(bool(v51))
        if wav_resolution not in VR_RES_TYPES:
            raise ValueError(f"wav_resolution {wav_resolution!r}: expected one of {sorted(VR_RES_TYPES)}")
        c.synth_res_type = VR_RES_TYPES[wav_resolution]
        conv = {None: 0, "mid_side": 1, "mid_side_c": 4, "stereo_n": 5}
        for d in range(1, nb + 1):
            bp = mp["band"][d]
            cc = bp.get("convert_channels") if v51 else None
            if cc not in conv:
                raise NotImplementedError(f"convert_channels {cc!r}")
            c.band[d - 1] = _VrBand(bp["sr"], bp["hl"], bp["n_fft"], bp["crop_start"], bp["crop_stop"], bp.get("hpf_start", 0),
                                    bp.get("hpf_stop", 0), bp.get("lpf_start", 0), bp.get("lpf_stop", 0), conv[cc],
                                    VR_RES_TYPES.get(bp.get("res_type", "polyphase"), 0))
        self._check(self._lib.asx_vr_begin(self._h, C.byref(c)))
        for name, t in state_dict.items():
            if hasattr(t, "detach"):
                t = t.detach().cpu().numpy()
            a = _f32(t).reshape(-1)
            self._check(self._lib.asx_net_set_tensor(self._h, name.encode(), _ptr(a), a.size))
        self._check(self._lib.asx_vr_commit(self._h))
        self.vr_bins = mp["bins"]
        self.vr_window = int(window_size)

    def resample_sinc(self, x: np.ndarray, ratio: float, mono_calls: bool = False) -> np.ndarray:
        """librosa.resample(x, orig_sr, target_sr, res_type="sinc_fastest") with ratio = float(target_sr) / orig_sr on x [C, n]
        (or [n]) -> [C, ceil(n * ratio)].  ``mono_calls``: channel by channel, as spec_utils.change_pitch_semitones calls it."""
        x = _f32(x)
        one = x.ndim == 1
        x2 = np.ascontiguousarray(x[None] if one else x)
        if x2.ndim != 2 or x2.shape[1] < 1:
            raise ValueError(f"resample_sinc expects [channels, n], got {x.shape}")
        n_out = int(np.ceil(x2.shape[1] * float(ratio)))
        y = np.empty((x2.shape[0], n_out), np.float32)
        self._check(self._lib.asx_resample_sinc(self._h, _ptr(x2), x2.shape[0], x2.shape[1], float(ratio), int(bool(mono_calls) or one),
                                                _ptr(y), n_out))
        return y[0] if one else y

    def resample_sinc_dev(self, x_ptr: int, channels: int, n_in: int, ratio: float, mono_calls: bool, y_ptr: int, n_out: int, stream: int = 0):
        self._check(self._lib.asx_resample_sinc_dev(self._h, x_ptr, int(channels), int(n_in), float(ratio), int(bool(mono_calls)), y_ptr,
                                                    int(n_out), stream or None))

    def vr_flops(self) -> float:
        return float(self._lib.asx_vr_flops(self._h))

    def vr_plan(self, n_samples: int):
        t, n = C.c_int32(), C.c_int64()
        self._check(self._lib.asx_vr_plan(self._h, int(n_samples), C.byref(t), C.byref(n)))
        return t.value, n.value

    def vr_forward(self, x: np.ndarray) -> np.ndarray:
        x = _f32(x)
        if x.ndim != 4 or x.shape[1] != 2 or x.shape[2] != self.vr_bins + 1 or x.shape[3] != self.vr_window:
            raise ValueError(f"expected [B, 2, {self.vr_bins + 1}, {self.vr_window}], got {x.shape}")
        out = np.empty_like(x)
        self._check(self._lib.asx_vr_forward(self._h, _ptr(x), x.shape[0], _ptr(out)))
        return out

    def vr_analysis(self, wave: np.ndarray) -> np.ndarray:
        """loading_mix: [2, n] -> complex64 [2, bins+1, n_frames] (transposed back to the reference layout)."""
        wave = _f32(wave)
        T, _ = self.vr_plan(wave.shape[1])
        buf = np.empty((2, T, self.vr_bins + 1, 2), np.float32)
        self._check(self._lib.asx_vr_analysis(self._h, _ptr(wave), wave.shape[1], _ptr(buf)))
        return np.ascontiguousarray(buf.view(np.complex64)[..., 0].transpose(0, 2, 1))

    def vr_separate(self, wave: np.ndarray, aggr_value: float, split_bin: int, is_non_accom: bool = False, aggr_correction=None,
                    enable_tta: bool = False, enable_post_process: bool = False, post_thres: float = 0.2,
                    high_end_process: bool = False):
        wave = _f32(wave)
        if wave.ndim != 2 or wave.shape[0] != 2:
            raise ValueError(f"Expected a 2-channel audio signal, but got shape {wave.shape}")
        _, n_out = self.vr_plan(wave.shape[1])
        corr = aggr_correction or {}
        pr = _VrParams(float(aggr_value), int(split_bin), int(bool(is_non_accom)), int(aggr_correction is not None),
                       float(corr.get("left", 0.0)), float(corr.get("right", 0.0)), int(bool(enable_tta)),
                       int(bool(enable_post_process)), float(post_thres), int(bool(high_end_process)))
        p = np.empty((2, n_out), np.float32)
        q = np.empty((2, n_out), np.float32)
        self._check(self._lib.asx_vr_separate(self._h, _ptr(wave), wave.shape[1], C.byref(pr), _ptr(p), _ptr(q)))
        return p, q

    def vr_separate_dev(self, wave_ptr: int, n_samples: int, primary_ptr: int, secondary_ptr: int, aggr_value: float,
                        split_bin: int, is_non_accom: bool = False, enable_tta: bool = False, enable_post_process: bool = False,
                        post_thres: float = 0.2, stream: int = 0, aggr_correction=None, high_end_process: bool = False):
        corr = aggr_correction or {}
        pr = _VrParams(float(aggr_value), int(split_bin), int(bool(is_non_accom)), int(aggr_correction is not None),
                       float(corr.get("left", 0.0)), float(corr.get("right", 0.0)), int(bool(enable_tta)),
                       int(bool(enable_post_process)), float(post_thres), int(bool(high_end_process)))
        self._check(self._lib.asx_vr_separate_dev(self._h, wave_ptr, n_samples, C.byref(pr), primary_ptr or None,
                                                  secondary_ptr or None, stream or None))

    # -- chunk-range halves of the sibling loops (multi-GPU sharding, sharding.py) ----------------------------
    def mdxc_chunks_dev(self, mix_ptr, n, overlap, k0, k1, out_ptr, stream=0):
        self._check(self._lib.asx_mdxc_chunks_dev(self._h, mix_ptr, n, int(overlap), k0, k1, out_ptr, stream or None))

    def mdxc_finalize_dev(self, chunks_ptr, n, overlap, out_ptr, stream=0):
        self._check(self._lib.asx_mdxc_finalize_dev(self._h, chunks_ptr, n, int(overlap), out_ptr, stream or None))

    def rof_plan(self, n, step):
        k, c = C.c_int32(), C.c_int64()
        self._check(self._lib.asx_rof_plan(self._h, n, int(step), C.byref(k), C.byref(c)))
        return {"n_chunks": k.value, "chunk_size": c.value}

    def rof_chunks_dev(self, mix_ptr, n, step, k0, k1, out_ptr, stream=0):
        self._check(self._lib.asx_rof_chunks_dev(self._h, mix_ptr, n, int(step), k0, k1, out_ptr, stream or None))

    def rof_finalize_dev(self, chunks_ptr, n, step, out_ptr, stream=0):
        self._check(self._lib.asx_rof_finalize_dev(self._h, chunks_ptr, n, int(step), out_ptr, stream or None))

    @staticmethod
    def _offs(shifts, offsets):
        return (C.c_int64 * shifts)(*[int(o) for o in offsets]) if shifts else None

    def ht_plan(self, n, shifts=0, offsets=None, overlap=0.25):
        k, c = C.c_int32(), C.c_int64()
        self._check(self._lib.asx_ht_plan(self._h, n, int(shifts), self._offs(shifts, offsets), float(overlap), C.byref(k), C.byref(c)))
        return {"n_chunks": k.value, "chunk_size": c.value}

    def ht_segments_dev(self, mix_ptr, n, k0, k1, out_ptr, shifts=0, offsets=None, overlap=0.25, flags=0, stream=0):
        self._check(self._lib.asx_ht_segments_dev(self._h, mix_ptr, n, int(shifts), self._offs(shifts, offsets), float(overlap), flags,
                                                  k0, k1, out_ptr, stream or None))

    def ht_fold_dev(self, mix_ptr, n, chunks_ptr, out_ptr, shifts=0, offsets=None, overlap=0.25, flags=0, stream=0):
        self._check(self._lib.asx_ht_fold_dev(self._h, mix_ptr, n, int(shifts), self._offs(shifts, offsets), float(overlap), flags,
                                              chunks_ptr, out_ptr, stream or None))

    def pcm16(self, stem: np.ndarray, max_peak: float = 1.0, min_peak=None):
        """write_audio_pydub's array work: stem [N, 2] (or [2, N] planar with planar=True semantics when shape[0] == 2)
        -> (int16 interleaved [N, 2], peak after normalisation)."""
        stem = np.asarray(stem, np.float32)
        planar = np.ascontiguousarray(stem.T if stem.shape[-1] == 2 and stem.shape[0] != 2 else stem)
        n = planar.shape[1]
        out = np.empty((n, 2), np.int16)
        pk = C.c_float()
        self._check(self._lib.asx_pcm16(self._h, _ptr(planar), n, float(max_peak), float(min_peak or 0.0), int(min_peak is not None),
                                        out.ctypes.data_as(C.POINTER(C.c_int16)), C.byref(pk)))
        return out, pk.value

    def pcm16_rows_dev(self, stem_ptr: int, n_samples: int, max_peak: float, min_peak, pcm_ptr: int, stream: int = 0,
                       want_peak: bool = True):
        """asx_pcm16_rows_dev: a device-resident [N, 2] stem -> int16 [N, 2] on the device; returns the peak after
        normalisation (synchronises the stream) or None."""
        pk = C.c_float()
        self._check(self._lib.asx_pcm16_rows_dev(self._h, stem_ptr, n_samples, float(max_peak), float(min_peak or 0.0),
                                                 int(min_peak is not None), pcm_ptr, C.byref(pk) if want_peak else None, stream or None))
        return pk.value if want_peak else None

    def pcm16_planar_dev(self, stem_ptr: int, n_samples: int, max_peak: float, min_peak, pcm_ptr: int, stream: int = 0,
                         want_peak: bool = True):
        """asx_pcm16_dev: a device-resident planar [2, N] stem -> int16 [N, 2] on the device; returns the peak after
        normalisation (synchronises the stream) or None."""
        pk = C.c_float()
        self._check(self._lib.asx_pcm16_dev(self._h, stem_ptr, n_samples, float(max_peak), float(min_peak or 0.0),
                                            int(min_peak is not None), pcm_ptr, C.byref(pk) if want_peak else None, stream or None))
        return pk.value if want_peak else None

    PCM_FORMATS = {"PCM_16": 16, "PCM_24": 24, "PCM_32": 32, "FLOAT": 0x120}

    def pcm_decode_dev(self, raw_ptr: int, frames: int, channels: int, subtype: str, mix_ptr: int, stream: int = 0,
                       want_peak: bool = True):
        """asx_pcm_decode_dev: a WAVE data chunk on the device -> float32 planar [2, frames]; returns max |mix| or None."""
        if subtype not in self.PCM_FORMATS:
            raise ValueError(f"no device decoder for subtype {subtype}")
        pk = C.c_float()
        self._check(self._lib.asx_pcm_decode_dev(self._h, raw_ptr, frames, channels, self.PCM_FORMATS[subtype], mix_ptr,
                                                 C.byref(pk) if want_peak else None, stream or None))
        return pk.value if want_peak else None

    def normalize(self, wave: np.ndarray, max_peak: float = 1.0, min_peak=None) -> np.ndarray:
        """spec_utils.normalize (uvr_lib_v5/spec_utils.py:99-115): scales ``wave`` IN PLACE when it is a C-contiguous float32
        array (like the reference, whose ``wave *= ...`` mutates the caller's array) and returns it."""
        a = wave if (isinstance(wave, np.ndarray) and wave.dtype == np.float32 and wave.flags.c_contiguous) else None
        buf = a if a is not None else np.ascontiguousarray(wave, np.float32)
        if buf.size:
            pk = C.c_float()
            self._check(self._lib.asx_normalize(self._h, _ptr(buf), buf.size, float(max_peak), float(min_peak or 0.0),
                                                int(min_peak is not None), C.byref(pk)))
        if a is None and isinstance(wave, np.ndarray) and wave.dtype == np.float32:
            wave[...] = buf                      # non-contiguous view (e.g. a transposed stem): write the result back
            return wave
        return buf

    def normalize_dev(self, wave_ptr: int, numel: int, max_peak: float = 1.0, min_peak=None, stream: int = 0):
        self._check(self._lib.asx_normalize_dev(self._h, wave_ptr, numel, float(max_peak), float(min_peak or 0.0),
                                                int(min_peak is not None), stream or None))

    def residual_dev(self, mix_ptr: int, stem_ptr: int, out_ptr: int, numel: int, stream: int = 0):
        self._check(self._lib.asx_residual_dev(self._h, mix_ptr, stem_ptr, out_ptr, numel, stream or None))

    ENSEMBLE_ALGORITHMS = ("avg_wave", "median_wave", "min_wave", "max_wave", "avg_fft", "median_fft", "min_fft", "max_fft",
                           "uvr_max_spec", "uvr_min_spec", "ensemble_wav")

    def ensemble(self, waveforms, algorithm: str = "avg_wave", weights=None) -> np.ndarray:
        """Ensembler.ensemble (ensembler.py:12-74): list of [2, N] waves (zero-padded to the longest) -> [2, N']."""
        if algorithm not in self.ENSEMBLE_ALGORITHMS:
            raise ValueError(f"Unknown ensemble algorithm: {algorithm}")
        ws = [_f32(w) for w in waveforms]
        if len(ws) == 1:
            return ws[0]
        if any(w.ndim != 2 or w.shape[0] != 2 for w in ws):
            raise ValueError("All waveforms must be stereo [2, N] for the accelerated ensemble")
        n = max(w.shape[1] for w in ws)
        stack = np.zeros((len(ws), 2, n), np.float32)
        for k, w in enumerate(ws):
            stack[k, :, : w.shape[1]] = w
        wt = None
        if weights is not None and len(weights) == len(ws) and np.all(np.isfinite(weights)) and np.sum(weights) != 0:
            wt = (C.c_double * len(ws))(*[float(v) for v in weights])
        out = np.empty((2, n), np.float32)
        n_out = C.c_int64()
        self._check(self._lib.asx_ensemble(self._h, _ptr(stack), len(ws), n, self.ENSEMBLE_ALGORITHMS.index(algorithm), wt, _ptr(out),
                                           C.byref(n_out)))
        return out.reshape(-1)[: 2 * n_out.value].reshape(2, n_out.value).copy()     # the engine writes planar [2, n_out]

    def invert_stem(self, mixture: np.ndarray, stem: np.ndarray) -> np.ndarray:
        """spec_utils.invert_stem(mixture [2, N], stem [2, N]) -> [N', 2]."""
        m, st = _f32(mixture), _f32(stem)
        n = min(m.shape[1], st.shape[1])
        m, st = np.ascontiguousarray(m[:, :n]), np.ascontiguousarray(st[:, :n])
        out = np.empty((2, n), np.float32)
        n_out = C.c_int64()
        self._check(self._lib.asx_invert_stem(self._h, _ptr(m), _ptr(st), n, _ptr(out), C.byref(n_out)))
        return np.ascontiguousarray(out.reshape(-1)[: 2 * n_out.value].reshape(2, n_out.value).T)

    def debug_fetch(self, name: str, shape) -> np.ndarray:
        out = np.empty(shape, np.float32)
        self._check(self._lib.asx_debug_fetch(self._h, name.encode(), _ptr(out), out.size))
        return out

    # -- plan ---------------------------------------------------------------
    def plan(self, n_samples: int, is_match_mix: bool = False) -> dict:
        p = _Plan()
        self._check(self._lib.asx_plan_query(self._h, n_samples, ASX_FLAG_MATCH_MIX if is_match_mix else 0,
                                             C.byref(p)))
        return {k: getattr(p, k) for k, _ in _Plan._fields_ if k != "reserved"}

    # -- the path -----------------------------------------------------------
    def demix(self, mix: np.ndarray, is_match_mix: bool = False) -> np.ndarray:
        mix = _f32(mix)
        if mix.ndim != 2 or mix.shape[0] != 2:
            raise ValueError(f"Expected a 2-channel audio signal, but got shape {mix.shape}")
        out = np.empty_like(mix)
        self._check(self._lib.asx_demix(self._h, _ptr(mix), mix.shape[1], _ptr(out),
                                        ASX_FLAG_MATCH_MIX if is_match_mix else 0))
        return out

    def demix_dev(self, mix_ptr: int, n_samples: int, out_ptr: int, is_match_mix: bool = False, stream: int = 0):
        self._check(self._lib.asx_demix_dev(self._h, mix_ptr, n_samples, out_ptr,
                                            ASX_FLAG_MATCH_MIX if is_match_mix else 0, stream or None))

    def demix_chunks_dev(self, mix_ptr: int, n_samples: int, k0: int, k1: int, chunk_out_ptr: int,
                         is_match_mix: bool = False, stream: int = 0):
        self._check(self._lib.asx_demix_chunks_dev(self._h, mix_ptr, n_samples, k0, k1, chunk_out_ptr,
                                                   ASX_FLAG_MATCH_MIX if is_match_mix else 0, stream or None))

    def finalize_dev(self, chunk_out_ptr: int, n_samples: int, out_ptr: int, is_match_mix: bool = False,
                     stream: int = 0):
        self._check(self._lib.asx_finalize_dev(self._h, chunk_out_ptr, n_samples, out_ptr,
                                               ASX_FLAG_MATCH_MIX if is_match_mix else 0, stream or None))

    def separate(self, mix: np.ndarray, max_peak: float, min_peak, compensate: float):
        """Stem algebra of MDXSeparator.separate on the device.  ``mix`` (float32 [2,N], C-contiguous) is
        normalised IN PLACE like the reference does; returns (primary [N,2], secondary [N,2])."""
        if mix.dtype != np.float32 or not mix.flags.c_contiguous or mix.ndim != 2 or mix.shape[0] != 2:
            raise ValueError("mix must be a C-contiguous float32 array of shape [2, N]")
        n = mix.shape[1]
        primary = np.empty((n, 2), np.float32)
        secondary = np.empty((n, 2), np.float32)
        self._check(self._lib.asx_separate(self._h, _ptr(mix), n, float(max_peak),
                                           float(min_peak) if min_peak is not None else 0.0,
                                           int(min_peak is not None), float(compensate), _ptr(primary),
                                           _ptr(secondary)))
        return primary, secondary

    def separate_dev(self, mix_ptr: int, n_samples: int, max_peak: float, min_peak, compensate: float,
                     primary_ptr: int, secondary_ptr: int, stream: int = 0):
        self._check(self._lib.asx_separate_dev(self._h, mix_ptr, n_samples, float(max_peak),
                                               float(min_peak) if min_peak is not None else 0.0,
                                               int(min_peak is not None), float(compensate), primary_ptr,
                                               secondary_ptr, stream or None))

    # -- stage hooks ----------------------------------------------------------
    def stft(self, wave: np.ndarray) -> np.ndarray:
        wave = _f32(wave)
        B, ch, Cn = wave.shape
        assert ch == 2
        T = Cn // self.cfg.hop_length + 1
        out = np.empty((B, 4, self.cfg.dim_f, T), np.float32)
        self._check(self._lib.asx_stft(self._h, _ptr(wave), B, Cn, _ptr(out)))
        return out

    def istft(self, spec: np.ndarray) -> np.ndarray:
        spec = _f32(spec)
        B, c4, Fq, T = spec.shape
        assert c4 == 4 and Fq == self.cfg.dim_f
        out = np.empty((B, 2, self.cfg.hop_length * (T - 1)), np.float32)
        self._check(self._lib.asx_istft(self._h, _ptr(spec), B, T, _ptr(out)))
        return out

    def net_forward(self, spec: np.ndarray) -> np.ndarray:
        spec = _f32(spec)
        out = np.empty_like(spec)
        self._check(self._lib.asx_net_forward(self._h, _ptr(spec), spec.shape[0], _ptr(out)))
        return out

    def run_model(self, wave: np.ndarray, is_match_mix: bool = False) -> np.ndarray:
        wave = _f32(wave)
        out = np.empty_like(wave)
        self._check(self._lib.asx_run_model(self._h, _ptr(wave), wave.shape[0], _ptr(out),
                                            ASX_FLAG_MATCH_MIX if is_match_mix else 0))
        return out

    def op_conv(self, op: str, x, w, b, aux=None, relu=True) -> np.ndarray:
        x, w, b = _f32(x), _f32(w), _f32(b)
        B, cin, t, f = x.shape
        if op == "up":
            cout = w.shape[1]
            shape = (B, cout, 2 * t, 2 * f)
        elif op == "down":
            cout = w.shape[0]
            shape = (B, cout, t // 2, f // 2)
        else:
            cout = w.shape[0]
            shape = (B, cout, t, f)
        aux = _f32(aux) if aux is not None else None
        y = np.empty(shape, np.float32)
        self._check(self._lib.asx_op_conv(self._h, op.encode(), _ptr(x), B, cin, t, f, _ptr(w), _ptr(b), cout,
                                          _optptr(aux), int(relu), _ptr(y)))
        return y

    def op_tdf(self, x, w, bias, scale, shift, res=None) -> np.ndarray:
        x, w, scale, shift = _f32(x), _f32(w), _f32(scale), _f32(shift)
        B, c, t, k = x.shape
        n = w.shape[0]
        bias = _f32(bias) if bias is not None else None
        res = _f32(res) if res is not None else None
        y = np.empty((B, c, t, n), np.float32)
        self._check(self._lib.asx_op_tdf(self._h, _ptr(x), B, c, t, k, _ptr(w), _optptr(bias), n, _ptr(scale),
                                         _ptr(shift), _optptr(res), _ptr(y)))
        return y

    def op_tdf_block(self, x, w0, scale0, shift0, w1, scale1, shift1, return_hidden=False):
        """x + tdf(x) of a TDF block with a bottleneck (two linears, launched as the net launches them); with return_hidden also the
        bottleneck activations as fp32 (decoded from the pair image when the first linear wrote one)."""
        x, w0, w1 = _f32(x), _f32(w0), _f32(w1)
        scale0, shift0, scale1, shift1 = _f32(scale0), _f32(shift0), _f32(scale1), _f32(shift1)
        B, c, t, f = x.shape
        n8 = w0.shape[0]
        if w0.shape != (n8, f) or w1.shape != (f, n8):
            raise AsxError(f"op_tdf_block: W0 {w0.shape} / W1 {w1.shape} do not fit rows of length {f}")
        y = np.empty((B, c, t, f), np.float32)
        h = np.empty((B, c, t, n8), np.float32) if return_hidden else None
        self._check(self._lib.asx_op_tdf_block(self._h, _ptr(x), B, c, t, f, _ptr(w0), _ptr(scale0), _ptr(shift0), n8, _ptr(w1),
                                               _ptr(scale1), _ptr(shift1), _ptr(y), _optptr(h)))
        return (y, h) if return_hidden else y

    # -- profiling --------------------------------------------------------------
    def profile_enable(self, on: bool = True):
        self._check(self._lib.asx_profile_enable(self._h, int(on)))

    def profile_launches(self):
        """Per-launch records of the current profile: [(class name, ms, flops, bytes)] in launch order."""
        n = C.c_int32()
        self._check(self._lib.asx_profile_launches(self._h, None, 0, C.byref(n)))
        recs = (_LaunchRec * max(1, n.value))()
        self._check(self._lib.asx_profile_launches(self._h, recs, n.value, C.byref(n)))
        return [(PROF_CLASSES[r.cls & 0xff], float(r.ms), float(r.flops), float(r.bytes)) for r in recs[: n.value]]

    def profile_launches_ex(self):
        """As profile_launches with a fifth field: the 16-bit MFMA products per multiply-add the launch executed -- 6 (bf16 x 6), 3 (fp16 x 3)
        or 0 (fp32 MFMA / VALU kernels) -- bits 8..15 of asx_launch_rec.cls (ABI 7)."""
        n = C.c_int32()
        self._check(self._lib.asx_profile_launches(self._h, None, 0, C.byref(n)))
        recs = (_LaunchRec * max(1, n.value))()
        self._check(self._lib.asx_profile_launches(self._h, recs, n.value, C.byref(n)))
        return [(PROF_CLASSES[r.cls & 0xff], float(r.ms), float(r.flops), float(r.bytes), (r.cls >> 8) & 0xff) for r in recs[: n.value]]

    def profile_read(self) -> dict:
        p = _Profile()
        self._check(self._lib.asx_profile_read(self._h, C.byref(p)))
        return {name: {"launches": int(p.launches[i]), "ms": float(p.ms[i]), "flops": float(p.flops[i]),
                       "bytes": float(p.bytes[i])} for i, name in enumerate(PROF_CLASSES)}
