"""DemucsSeparator on the HIP engine: drop-in for audio_separator/separator/architectures/demucs_separator.py.

Same constructor, ``separate`` / ``demix_demucs`` contract, source maps and output naming.  The model package
(``.th`` + bag ``.yaml``, read without importing or executing any Demucs code -- model_files.py) is loaded once and
kept resident instead of being re-read for every file (:119-124); ``apply_model`` with its shift trick, segment split
and triangular fold, the HTDemucs / HDemucs forward and the standardise / de-standardise / stem swap of ``demix_demucs``
(:162-194) are ``asx_ht_demix`` / ``asx_hd_demix``.
"""
from __future__ import annotations


import numpy as np

from ..common_separator import CommonSeparator
from ..demucs import DemucsDemixer

DEMUCS_4_SOURCE = ["drums", "bass", "other", "vocals"]
DEMUCS_2_SOURCE_MAPPER = {CommonSeparator.INST_STEM: 0, CommonSeparator.VOCAL_STEM: 1}
DEMUCS_4_SOURCE_MAPPER = {CommonSeparator.BASS_STEM: 0, CommonSeparator.DRUM_STEM: 1, CommonSeparator.OTHER_STEM: 2,
                          CommonSeparator.VOCAL_STEM: 3}
DEMUCS_6_SOURCE_MAPPER = {CommonSeparator.BASS_STEM: 0, CommonSeparator.DRUM_STEM: 1, CommonSeparator.OTHER_STEM: 2,
                          CommonSeparator.VOCAL_STEM: 3, CommonSeparator.GUITAR_STEM: 4, CommonSeparator.PIANO_STEM: 5}


class DemucsSeparator(CommonSeparator):
    def __init__(self, common_config, arch_config):
        super().__init__(config=common_config)
        self._read_options(arch_config, (("segment_size", "Default"), ("shifts", 2), ("overlap", 0.25), ("segments_enabled", True)))
        self.logger.debug(f"Demucs arch params: segment_size={self.segment_size}, segments_enabled={self.segments_enabled}, "
                          f"shifts={self.shifts}, overlap={self.overlap}")
        self.demucs_source_map = DEMUCS_4_SOURCE_MAPPER
        self.demucs_model_instance = None
        self._common, self._arch = dict(common_config), dict(arch_config)
        self._max_batch = int(arch_config.get("asx_max_batch", 0))
        self.logger.info("Demucs plugin ready (the model package is read at the first separate())")

    def load_model(self):
        """demucs_separator.py:119-124 (get_demucs_model + demucs_segments + .to(device).eval()), done once."""
        if self.demucs_model_instance is None:
            common = dict(self._common)
            common["logger"] = self.logger
            self.demucs_model_instance = DemucsDemixer(common, self._arch, models=common.get("asx_models"),
                                                       weights=common.get("asx_weights"), max_batch=self._max_batch)
            self.demucs_model_instance._load(0)
            self.engine = self.demucs_model_instance.engine
        return self.demucs_model_instance

    def demix_demucs(self, mix):
        """demucs_separator.py:162-194: [2, N] -> [S, 2, N] with sources 0 / 1 swapped."""
        dm = self.load_model()
        dm.shifts, dm.overlap, dm.segments_enabled = self.shifts, self.overlap, self.segments_enabled
        out = dm.demix(np.ascontiguousarray(mix, np.float32))
        self.engine = dm.engine
        return out

    def _demix_on_device(self):
        """RIFF/WAVE input at the model's rate: data chunk -> pinned -> HBM -> asx_pcm_decode_dev -> the demix (single model or
        bag) -> stems [S, 2, N] that STAY in HBM; ``source`` is their pinned host mirror and every ``source[i].T`` view handed to
        write_audio is registered against its device tensor, so the int16 pass runs on the device (asx_pcm16_dev) without a
        second upload.  (None, None) when the file needs the host decoder or the configuration the host combine."""
        dm = self.load_model()                      # binds self.engine, which the device decode needs
        mix_d = self._device_mix(self.audio_file_path)
        if mix_d is None:
            return None, None
        t0 = self._now()
        dm.shifts, dm.overlap, dm.segments_enabled = self.shifts, self.overlap, self.segments_enabled
        out_d = dm.demix_dev(mix_d) if hasattr(dm, "demix_dev") else None
        self.engine = dm.engine
        if out_d is None:
            return None, None
        t0 = self._tick("demix", t0)
        source, views = self._host_planar_stems(out_d)
        self._sync()
        self._tick("stems_d2h", t0)
        return source, views

    def separate(self, audio_file_path, custom_output_names=None):
        """demucs_separator.py:83-160."""
        self._begin_file(audio_file_path)
        source, views = self._demix_on_device()
        if source is None:
            mix = self.prepare_mix(self.audio_file_path)
            self.load_model()
            source = self.demix_demucs(mix)
            self.clear_gpu_cache()

        n = len(source)
        self.demucs_source_map = {2: DEMUCS_2_SOURCE_MAPPER, 6: DEMUCS_6_SOURCE_MAPPER}.get(n, DEMUCS_4_SOURCE_MAPPER)
        files = []
        for stem_name, index in self.demucs_source_map.items():
            if self.output_single_stem is not None and stem_name.lower() != self.output_single_stem.lower():
                self.logger.debug(f"{stem_name}: not written (output_single_stem = {self.output_single_stem})")
                continue
            path = self.get_stem_output_path(stem_name, custom_output_names)
            self.final_process(path, views[index] if views is not None else source[index].T, stem_name)
            files.append(path)
        return files
