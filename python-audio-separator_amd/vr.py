"""Host-side mirror of the reference VR plugin's array path (architectures/vr_separator.py).

``VRDemixer(common_config, arch_config, state_dict, nn_arch_size)`` follows ``VRSeparator.__init__`` (:27-113) for the
configuration (model_data -> vr_model_param JSON, enable_tta, enable_post_process, post_process_threshold, batch_size,
window_size, aggression) and ``separate`` (:115-253) for the arrays: ``separate_stems(wave [2, n]) -> (primary [n', 2],
secondary [n', 2])`` where wave is what ``librosa.load(file, sr=band[N].sr, mono=False)`` returns (decode stays with the
reference).  Multiband analysis, patching, the CascadedASPPNet, mask post-processing and the multiband synthesis all run
in libasx.so (asx_vr_separate).

Resampling between bands follows the reference's choices (``resolve_res_type``): the synthesis chain runs ``sinc_fastest``
everywhere except macOS on ARM, where it runs ``polyphase`` (uvr_lib_v5/spec_utils.py:33-38); the analysis chain runs each
band's own ``res_type`` (vr_separator.py:267,282).  ``arch_config["asx_res_type"]`` = "polyphase" | "sinc_fastest" overrides
the platform rule (the golden vectors were written through a polyphase stand-in for librosa and say so).  VR 5.1
checkpoints (nets_new.CascadedNet: model_data with nout / nout_lstm, or file sizes 56817 / 218409) take the same path
with the is_v51_model branches.
"""
from __future__ import annotations

import json
import os

import numpy as np

from .engine import Engine, MDXConfig
from .mdx import _device_index

NON_ACCOM_STEMS = ("Vocals", "Other", "Bass", "Drums", "Guitar", "Piano", "Synthesizer", "Strings", "Woodwinds", "Brass",
                   "Wind Inst")   # common_separator.py:46
NN_ARCH_SIZES = [31191, 33966, 56817, 123821, 123812, 129605, 218409, 537238, 537227]
VR_5_1 = (56817, 218409)


def reference_wav_resolution() -> str:
    """spec_utils.py:33-38: the synthesis res_type the reference picks on this platform."""
    import platform
    if platform.system() == "Darwin":
        arm = "arm" in platform.processor().lower() or "arm" in platform.platform().lower()
        return "polyphase" if arm else "sinc_fastest"
    return "sinc_fastest"


def resolve_res_type(asked) -> str:
    """``arch_config["asx_res_type"]``: None / "auto" -> the reference's platform rule; "sinc" is short for "sinc_fastest"."""
    if asked in (None, "auto"):
        return reference_wav_resolution()
    if asked == "sinc":
        return "sinc_fastest"
    if asked not in ("polyphase", "sinc_fastest"):
        raise ValueError(f"asx_res_type {asked!r}: expected 'polyphase', 'sinc_fastest' or 'auto'")
    return asked


def load_model_params(path_or_dict) -> dict:
    """ModelParameters.__init__ (uvr_lib_v5/vr_network/model_param_init.py:48-71)."""
    if isinstance(path_or_dict, dict):
        p = dict(path_or_dict)
    else:
        with open(path_or_dict, "r") as f:
            p = json.loads(f.read(), object_pairs_hook=lambda kv: {(int(k) if k.isdigit() else k): v for k, v in kv})
    for k in ("mid_side", "mid_side_b", "mid_side_b2", "stereo_w", "stereo_n", "reverse"):
        p.setdefault(k, False)
    if "n_bins" in p:
        p["bins"] = p["n_bins"]
    return p


def model_capacity(nn_architecture: int):
    """nets.determine_model_capacity (nets.py:65-93)."""
    if nn_architecture in (31191, 33966, 129605):
        return [(2, 16), (2, 16), (18, 8, 1, 1, 0), (8, 16), (34, 16, 1, 1, 0), (16, 32), (32, 2, 1), (16, 2, 1), (16, 2, 1)]
    if nn_architecture in (123821, 123812):
        return [(2, 32), (2, 32), (34, 16, 1, 1, 0), (16, 32), (66, 32, 1, 1, 0), (32, 64), (64, 2, 1), (32, 2, 1), (32, 2, 1)]
    if nn_architecture in (537238, 537227):
        return [(2, 64), (2, 64), (66, 32, 1, 1, 0), (32, 64), (130, 64, 1, 1, 0), (64, 128), (128, 2, 1), (64, 2, 1), (64, 2, 1)]
    raise NotImplementedError(f"VR architecture size {nn_architecture} is not a CascadedASPPNet capacity table entry (VR 5.1 sizes take nout / nout_lstm instead)")


def reference_params_dir():
    """Directory of the reference's VR model-parameter JSON files (uvr_lib_v5/vr_network/modelparams, the path
    VRSeparator.__init__ builds at vr_separator.py:46-50).  The files are data of the installed ``audio_separator``
    package this plugin is loaded into; they are looked up there, not copied.  None when the package is absent."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("audio_separator")
    except (ImportError, ValueError):
        spec = None
    for root in (list(spec.submodule_search_locations) if spec and spec.submodule_search_locations else []):
        d = os.path.join(root, "separator", "uvr_lib_v5", "vr_network", "modelparams")
        if os.path.isdir(d):
            return d
    return os.environ.get("ASX_VR_PARAMS_DIR")


def nn_arch_size_from_file(model_path: str) -> int:
    """vr_separator.py:161-164: nearest known size to ceil(file bytes / 1024)."""
    import math
    model_size = math.ceil(os.stat(model_path).st_size / 1024)
    return min(NN_ARCH_SIZES, key=lambda x: abs(x - model_size))


class VRDemixer:
    def __init__(self, common_config: dict, arch_config: dict, state_dict: dict, nn_arch_size: int, capacity=None,
                 offset: int | None = None, max_batch: int = 0):
        md = common_config.get("model_data", {})
        # vr_separator.py:38-43,161-170: VR 5.1 = model_data carries nout / nout_lstm, or the file size says so
        self.model_capacity = (32, 128)
        self.is_vr_51_model = False
        if "nout" in md and "nout_lstm" in md:
            self.model_capacity = (md["nout"], md["nout_lstm"])
            self.is_vr_51_model = True
        if nn_arch_size in VR_5_1:
            self.is_vr_51_model = True
        if offset is None:
            offset = 64 if self.is_vr_51_model else 128        # nets_new.py:101 / nets.py:130
        mp = common_config.get("model_params")
        if mp is None:
            here = common_config.get("vr_params_dir") or reference_params_dir() or ""
            mp = os.path.join(here, f"{md['vr_model_param']}.json")
        self.model_params = load_model_params(mp)
        self.primary_stem_name = common_config.get("primary_stem_name", md.get("primary_stem", "Instrumental"))
        self.enable_tta = arch_config.get("enable_tta", False)
        self.enable_post_process = arch_config.get("enable_post_process", False)
        self.post_process_threshold = arch_config.get("post_process_threshold", 0.2)
        self.batch_size = arch_config.get("batch_size", 1)
        self.window_size = arch_config.get("window_size", 512)
        self.high_end_process = bool(arch_config.get("high_end_process", False))
        self.wav_resolution = resolve_res_type(arch_config.get("asx_res_type"))
        self.aggression = float(int(arch_config.get("aggression", 5)) / 100)
        self.aggressiveness = {"value": self.aggression, "split_bin": self.model_params["band"][1]["crop_stop"],
                               "aggr_correction": self.model_params.get("aggr_correction")}
        self.model_samplerate = self.model_params["sr"]
        bins = self.model_params["bins"]
        self.engine = Engine(MDXConfig(n_fft=2 * bins, hop_length=bins // 2, dim_f=bins, segment_size=8),
                             device=_device_index(common_config.get("torch_device", 0)))
        self.engine.load_vr(self.model_params, nn_arch_size,
                            None if self.is_vr_51_model else (capacity or model_capacity(nn_arch_size)), state_dict,
                            window_size=self.window_size, offset=offset, max_batch=self._patches_per_pass(max_batch, common_config),
                            v51=self.model_capacity if self.is_vr_51_model else None, wav_resolution=self.wav_resolution)

    # patches per net pass -- an engine knob, results do not depend on it (engine_vr.h vr_mask_pass).  The workspace costs about
    # 1 GB per patch on the 4band_44100 layout (activations of the cascade at window 512), so:
    #   * an explicit ``asx_max_batch`` (arch config) / ``max_batch`` argument is honoured as given;
    #   * otherwise DEFAULT_PATCHES (48: the measured-best pass size on a 288 GB MI355X), clamped to what fits in 60 % of the HBM
    #     that is free right now, and never below the user's ``batch_size`` clamp of 1.
    DEFAULT_PATCHES = 48
    BYTES_PER_PATCH = 1.1e9

    def _patches_per_pass(self, explicit: int, common_config: dict) -> int:
        if explicit and explicit > 0:
            return int(explicit)
        want = self.DEFAULT_PATCHES
        try:
            import torch
            free, _total = torch.cuda.mem_get_info(_device_index(common_config.get("torch_device", 0)))
            per = self.BYTES_PER_PATCH * (self.window_size / 512.0)
            want = max(1, min(want, int(0.6 * free / per)))
        except Exception:          # no torch / no device query: keep the default, the engine reports an allocation failure itself
            pass
        return want

    def separate_stems_dev(self, wave_d):
        """The same with the wave [2, n] and both stems in HBM: returns one CUDA tensor [2 (primary, secondary), 2, n_out]
        (planar stems; ``stems[i].T`` is what the reference hands to write_audio)."""
        import torch
        n = wave_d.shape[1]
        _, n_out = self.engine.vr_plan(n)
        out = torch.empty((2, 2, n_out), dtype=torch.float32, device=wave_d.device)
        self.engine.vr_separate_dev(wave_d.data_ptr(), n, out[0].data_ptr(), out[1].data_ptr(), self.aggressiveness["value"],
                                    self.aggressiveness["split_bin"], is_non_accom=self.primary_stem_name in NON_ACCOM_STEMS,
                                    enable_tta=self.enable_tta, enable_post_process=self.enable_post_process,
                                    post_thres=self.post_process_threshold,
                                    stream=torch.cuda.current_stream(wave_d.device).cuda_stream,
                                    aggr_correction=self.aggressiveness["aggr_correction"], high_end_process=self.high_end_process)
        return out

    def separate_stems(self, wave: np.ndarray, want_primary: bool = True, want_secondary: bool = True):
        """(primary_source, secondary_source) as [n', 2] arrays (vr_separator.py:211-236, before final_process), at the
        MODEL's sample rate ``self.model_samplerate``: for parameter sets with sr != 44100 the reference then calls
        librosa.resample(..., target_sr=44100) (:218-220, :238-240), which the file-level VRSeparator does on the host.
        A stem that is not wanted (output_single_stem) comes back as None."""
        p, s = self.engine.vr_separate(wave, self.aggressiveness["value"], self.aggressiveness["split_bin"],
                                       is_non_accom=self.primary_stem_name in NON_ACCOM_STEMS,
                                       aggr_correction=self.aggressiveness["aggr_correction"], enable_tta=self.enable_tta,
                                       enable_post_process=self.enable_post_process, post_thres=self.post_process_threshold,
                                       high_end_process=self.high_end_process)
        return (p.T if want_primary else None), (s.T if want_secondary else None)
