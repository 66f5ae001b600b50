"""VRSeparator on the HIP engine: drop-in for audio_separator/separator/architectures/vr_separator.py.

Same constructor, attributes and ``separate`` contract (:115-253).  ``loading_mix`` / ``inference_vr`` / ``spec_to_wav``
and the spec_utils functions under them are one C call (``asx_vr_separate``, vr.py:VRDemixer); the network is built and
the ``.pth`` read once, at the first file, instead of at every ``separate`` (:158-178).

Resampler (vr.py ``resolve_res_type``): like the reference, the *synthesis* chain runs libsamplerate's ``sinc_fastest``
everywhere except macOS on ARM, where it runs ``polyphase`` (uvr_lib_v5/spec_utils.py:33-38), and the analysis chain runs each
band's own ``res_type``.  ``sinc_fastest`` is the library's published algorithm on a regenerated Kaiser-sinc table -- the
library's own table cannot be had here (INTEGRATION.md "VR resampler": parity unpinned for this converter, measured bounds
there); ``arch_config["asx_res_type"]`` = "polyphase" | "sinc_fastest" overrides the platform rule.  A warning is logged once
when a band asks for a converter the engine serves with the polyphase filter instead (``sinc_medium``, ``sinc_best``,
``kaiser_*`` -- in the shipped parameter files only top bands carry those, where they matter for non-44.1 kHz input files
decoded on the host).
"""
from __future__ import annotations

import math
import os

import numpy as np

from .. import audio_io
from ..common_separator import CommonSeparator
from ..model_files import read_state_dict
from ..vr import NN_ARCH_SIZES, VR_5_1, VRDemixer, load_model_params, reference_params_dir, reference_wav_resolution, resolve_res_type  # noqa: F401


class VRSeparator(CommonSeparator):
    def __init__(self, common_config, arch_config: dict):
        super().__init__(config=common_config)
        self.model_capacity = 32, 128
        self.is_vr_51_model = False
        if "nout" in self.model_data.keys() and "nout_lstm" in self.model_data.keys():
            self.model_capacity = self.model_data["nout"], self.model_data["nout_lstm"]
            self.is_vr_51_model = True
        params_dir = common_config.get("vr_params_dir") or reference_params_dir()
        self.model_params_path = os.path.join(params_dir or "", f"{self.model_data['vr_model_param']}.json")
        self.model_params = load_model_params(self.model_params_path)

        self._read_options(arch_config, (("enable_tta", False), ("enable_post_process", False), ("post_process_threshold", 0.2),
                                         ("batch_size", 1), ("window_size", 512), ("high_end_process", False)))
        self.input_high_end_h = None
        self.input_high_end = None
        self.aggression = float(int(arch_config.get("aggression", 5)) / 100)
        self.aggressiveness = {"value": self.aggression, "split_bin": self.model_params["band"][1]["crop_stop"],
                               "aggr_correction": self.model_params.get("aggr_correction")}
        self.model_samplerate = self.model_params["sr"]
        self.res_type = resolve_res_type(arch_config.get("asx_res_type"))
        self._common, self._arch = dict(common_config), dict(arch_config)
        self._dm = None
        self.model_run = None
        self.logger.debug(f"VR arch params: enable_tta={self.enable_tta}, enable_post_process={self.enable_post_process}, "
                          f"post_process_threshold={self.post_process_threshold}, batch_size={self.batch_size}, "
                          f"window_size={self.window_size}, high_end_process={self.high_end_process}, aggression={self.aggression}")
        self.logger.info(f"VR plugin ready ({self.model_data['vr_model_param']}, {'5.1' if self.is_vr_51_model else '5.0'} net on the HIP engine)")

    def _warn_resampler(self):
        bands = self.model_params["band"]
        lower = {str(bands[d].get("res_type")) for d in bands if d != max(bands)}     # converters loading_mix really runs
        foreign = sorted(w for w in lower if w not in ("polyphase", "sinc_fastest", "None"))
        if foreign:
            self.logger.warning(f"VR analysis resampling: band entries ask for {foreign} (libsamplerate / resampy); this engine "
                                "serves them with the polyphase filter (INTEGRATION.md, 'VR resampler').")

    def load_model(self):
        """vr_separator.py:158-178: architecture size from the file size, CascadedASPPNet / CascadedNet, load_state_dict."""
        if self._dm is not None:
            return self._dm
        state_dict = self._common.get("asx_state_dict")
        if state_dict is None:
            model_size = math.ceil(os.stat(self.model_path).st_size / 1024)
            nn_arch_size = min(NN_ARCH_SIZES, key=lambda x: abs(x - model_size))
            state_dict = read_state_dict(self.model_path)
        else:
            nn_arch_size = self._common["asx_nn_arch_size"]
        nn_arch_size = self._common.get("asx_nn_arch_size", nn_arch_size)
        if nn_arch_size in VR_5_1 or self.is_vr_51_model:
            self.is_vr_51_model = True
        common = dict(self._common)
        common.update(model_params=self.model_params, primary_stem_name=self.primary_stem_name, logger=self.logger)
        self._dm = VRDemixer(common, self._arch, state_dict, nn_arch_size, capacity=self._common.get("asx_capacity"),
                             max_batch=int(self._arch.get("asx_max_batch", 0)))
        self.engine = self._dm.engine
        self.model_run = self._dm.engine.vr_forward
        self._warn_resampler()
        return self._dm

    def separate(self, audio_file_path, custom_output_names=None):
        """vr_separator.py:115-253."""
        self._reset_file_state()
        self._begin_file(audio_file_path)
        try:
            self.input_audio_subtype = audio_io.info(audio_file_path)["subtype"]
            if "24" in self.input_audio_subtype:
                self.wav_subtype, self.input_bit_depth = "PCM_24", 24
            elif "32" in self.input_audio_subtype:
                self.wav_subtype, self.input_bit_depth = "PCM_32", 32
            else:
                self.wav_subtype, self.input_bit_depth = "PCM_16", 16
        except Exception as e:
            self.logger.warning(f"{audio_file_path}: no container info ({e}); stems will be written as PCM_16")
            self.wav_subtype, self.input_audio_subtype, self.input_bit_depth = "PCM_16", None, 16
        # the reference's VR path never goes through prepare_mix, so input_subtype stays None and the soundfile writer picks
        # PCM_16 / 24 / 32 from input_bit_depth (common_separator.py:390-402); keep that
        self.input_subtype = None

        dm = self.load_model()
        bands = self.model_params["band"]
        top = bands[len(bands)]

        if self.output_single_stem and self.output_single_stem.lower() not in (self.primary_stem_name.lower(),
                                                                               self.secondary_stem_name.lower()):
            self.logger.warning(f"output_single_stem = '{self.output_single_stem}' names neither '{self.primary_stem_name}' nor "
                                f"'{self.secondary_stem_name}' (model {self.model_name}): ignored, both stems are written")
            self.output_single_stem = None
        want_p, want_s = self._wanted(self.primary_stem_name), self._wanted(self.secondary_stem_name)

        primary = secondary = None
        # device-resident path (RIFF/WAVE at the top band's rate, which is also the rate the stems are written at): the data
        # chunk is decoded on the device and both stems stay in HBM until the writer's int16 pass
        keep = (self.input_subtype, self.input_bit_depth)
        wave_d = self._device_mix(audio_file_path, check_silent=False) if (top["sr"] == self.sample_rate and self.model_samplerate == 44100) else None
        self.input_subtype, self.input_bit_depth = keep          # _device_mix records prepare_mix's fields; VR keeps its own (above)
        if wave_d is not None:
            t0 = self._now()
            stems_d = dm.separate_stems_dev(wave_d)
            t0 = self._tick("demix", t0)
            _, views = self._host_planar_stems(stems_d)
            self._sync()
            self._tick("stems_d2h", t0)
            primary, secondary = (views[0] if want_p else None), (views[1] if want_s else None)
        else:
            # loading_mix (:255-291): the top band is the file decoded at the band's rate; everything below happens on the device
            wave, _ = audio_io.load(audio_file_path, sr=top["sr"], mono=False)
            if wave.ndim == 1:
                wave = np.asarray([wave, wave])
            wave = np.ascontiguousarray(wave, np.float32)
            primary, secondary = dm.separate_stems(wave, want_primary=want_p, want_secondary=want_s)

        # primary first here (vr_separator.py:211-246), unlike the MDX family
        files = []
        if want_p:
            if not isinstance(self.primary_source, np.ndarray):
                self.primary_source = self._to_44100(primary)
            self.primary_stem_output_path = self._emit_stem(self.primary_stem_name, self.primary_source, custom_output_names, files)
        if want_s:
            if not isinstance(self.secondary_source, np.ndarray):
                self.secondary_source = self._to_44100(secondary)
            self.secondary_stem_output_path = self._emit_stem(self.secondary_stem_name, self.secondary_source, custom_output_names, files)
        return files

    def _to_44100(self, stem):
        """vr_separator.py:218-220, :238-240: models trained at another rate are brought back with librosa.resample's
        default converter (soxr_hq) -- a host library the reference depends on; it is used when present."""
        if self.model_samplerate == 44100:
            return stem
        librosa = audio_io._optional("librosa")
        if librosa is not None and hasattr(librosa, "resample"):
            return librosa.resample(stem.T, orig_sr=self.model_samplerate, target_sr=44100).T
        # no librosa in this environment (the reference itself could not run here): the engine's libsamplerate-style converter
        # stands in for soxr_hq -- same band limit to well below 1e-3, not the same filter
        if not getattr(self, "_warned_soxr", False):
            self._warned_soxr = True
            self.logger.warning(f"model sample rate {self.model_samplerate} != 44100 and librosa is not installed: the final resample "
                                "uses the engine's sinc_fastest converter instead of librosa's default soxr_hq")
        ratio = float(44100) / self.model_samplerate
        return np.ascontiguousarray(self.engine.resample_sinc(np.ascontiguousarray(np.asarray(stem, np.float32).T), ratio).T)
