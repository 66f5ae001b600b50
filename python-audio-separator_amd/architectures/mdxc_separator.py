"""MDXCSeparator on the HIP engine: drop-in for audio_separator/separator/architectures/mdxc_separator.py.

TFC-TDF v3 (MDX23C) checkpoints and BS / Mel-Band Roformer checkpoints; same constructor, ``load_model`` / ``separate`` /
``demix`` contract, stem dictionary, file naming and the "shorter than 10 s -> override_model_segment_size" rule
(:131-138).  ``torch.load`` + ``load_state_dict`` (:76-116) become ``asx_v3_*`` / ``asx_rof_*``; ``spec_utils.normalize`` of
the mix and of every stem (:147, :170-190) is ``asx_normalize``; the chunk loops are ``asx_mdxc_demix`` / ``asx_rof_demix``.
"""
from __future__ import annotations


import numpy as np

from ..common_separator import CommonSeparator
from ..mdxc import MDXCDemixer


class MDXCSeparator(CommonSeparator):
    def __init__(self, common_config, arch_config):
        super().__init__(config=common_config)
        self._read_options(arch_config, (("segment_size", 256), ("override_model_segment_size", False), ("overlap", 8), ("batch_size", 1),
                                         ("pitch_shift", 0), ("process_all_stems", True)))
        self.logger.debug(f"MDXC arch params: batch_size={self.batch_size}, segment_size={self.segment_size}, overlap={self.overlap}, "
                          f"override_model_segment_size={self.override_model_segment_size}, pitch_shift={self.pitch_shift}")
        self.is_roformer = getattr(self, "is_roformer_model", False)
        self._common, self._arch = dict(common_config), dict(arch_config)
        self._max_batch = int(arch_config.get("asx_max_batch", 0))
        self._demixers = {}                 # one engine per chunk geometry (override_model_segment_size may flip per file)

        self.load_model()

        self._reset_file_state()
        training = self.model_data.get("training", {}) or {}
        self.is_primary_stem_main_target = bool(training.get("target_instrument"))
        self.logger.info(f"MDXC model ready ({'Roformer' if self.is_roformer else 'TFC-TDF v3'} on the HIP engine)")

    # ---- weights ---------------------------------------------------------------
    def _demixer(self) -> MDXCDemixer:
        key = bool(self.override_model_segment_size)
        dm = self._demixers.get(key)
        if dm is None:
            common = dict(self._common)
            common["logger"] = self.logger
            common["primary_stem_name"], common["secondary_stem_name"] = self.primary_stem_name, self.secondary_stem_name
            arch = dict(self._arch)
            arch["override_model_segment_size"] = key
            dm = MDXCDemixer(common, arch, state_dict=self._state_dict, max_batch=self._max_batch)
            if self.roformer_loader is not None and getattr(dm, "roformer_loader", None) is not None:
                self.roformer_loader._loading_stats = dm.roformer_loader.get_loading_stats()
            self._demixers[key] = dm
        dm.overlap = self.overlap
        self.engine = dm.engine
        return dm

    def load_model(self):
        """mdxc_separator.py:76-116: the checkpoint is read once; a failing / corrupt file exits like the reference."""
        import sys
        from ..model_files import read_state_dict
        from ..roformer_config import read_checkpoint
        self._state_dict = self._common.get("asx_state_dict")
        try:
            if self._state_dict is None:
                self._state_dict = read_checkpoint(self.model_path) if self.is_roformer else read_state_dict(self.model_path)
            self._demixer()
        except RuntimeError as e:
            # same outcome as the reference (mdxc_separator.py:108-116): a checkpoint that cannot be read ends the process
            self.logger.error(f"{self.model_path}: the checkpoint could not be loaded ({e}); the file is probably truncated or "
                              "corrupt -- delete it so that it is fetched again")
            sys.exit(1)

    # ---- the path ----------------------------------------------------------------
    def demix(self, mix: np.ndarray):
        """mdxc_separator.py:257-468: dict of stems, or the primary array for a single-target model without residual."""
        return self._demixer().demix(mix)

    def _separate_on_device(self, custom_output_names):
        """The same steps with every array in HBM (RIFF/WAVE input at the model's rate): decode on the device, normalise the mix in
        place (asx_normalize_dev), demix, residual stem, normalise every stem in place, host mirrors (pinned) for
        ``primary_source`` / ``secondary_source``, int16 pass on the device per written stem.  None: take the generic path."""
        if self.pitch_shift != 0:
            return None               # the pitch round trip runs through demix() on host arrays (mdxc.py _demix_pitched)
        if self.engine is None:
            self._demixer()
        mix_d = self._device_mix(self.audio_file_path)
        if mix_d is None:
            return None
        n = mix_d.shape[1]
        seconds = n / self.sample_rate
        if seconds < 10.0 and not self.override_model_segment_size:
            self.override_model_segment_size = True
            self.logger.warning(f"{seconds:.2f} s of audio (< 10 s): switching to the configured segment size "
                                "(override_model_segment_size), as the reference does for short files")
        dm = self._demixer()
        if dm.engine.device != mix_d.device.index:
            return None
        t0 = self._now()
        eng, st = dm.engine, self._stream()
        thr, amp = self.normalization_threshold, self.amplification_threshold
        eng.normalize_dev(mix_d.data_ptr(), 2 * n, thr, amp, stream=st)
        names, stems_d = dm.demix_dev(mix_d)
        training = self.model_data.get("training", {}) or {}
        single = names == [None]
        if single:
            wanted = [0]
        else:
            order = [training["target_instrument"]] if training.get("target_instrument") else list(training.get("instruments") or [])
            if self.process_all_stems and len(order) > 2:
                wanted = [names.index(k) for k in order]
            else:
                wanted = [names.index(self.primary_stem_name), names.index(self.secondary_stem_name)]
            for i in sorted(set(wanted)):           # norm(source[name]) of the reference, once per stem, in place
                eng.normalize_dev(stems_d[i].data_ptr(), 2 * n, thr, amp, stream=st)
        t0 = self._tick("demix", t0)
        _, views = self._host_planar_stems(stems_d)
        self._sync()
        self._tick("stems_d2h", t0)
        files = []
        if single:
            if self._wanted(self.primary_stem_name):
                if not isinstance(self.primary_source, np.ndarray):
                    self.primary_source = views[0]
                self.primary_stem_output_path = self._emit_stem(self.primary_stem_name, self.primary_source, custom_output_names, files)
            return files
        if self.process_all_stems and len(wanted) > 2:
            for k, i in zip(order, wanted):
                self._emit_stem(k, views[i], custom_output_names, files)
            return files
        if not isinstance(self.primary_source, np.ndarray):
            self.primary_source = views[wanted[0]]
        if not isinstance(self.secondary_source, np.ndarray):
            self.secondary_source = views[wanted[1]]
        return self._emit_pair(custom_output_names)

    def separate(self, audio_file_path, custom_output_names=None):
        """mdxc_separator.py:118-227."""
        self._reset_file_state()
        self._begin_file(audio_file_path)
        files = self._separate_on_device(custom_output_names)
        if files is not None:
            return files
        mix = self.prepare_mix(self.audio_file_path)

        seconds = mix.shape[1] / self.sample_rate
        if seconds < 10.0 and not self.override_model_segment_size:
            self.override_model_segment_size = True
            self.logger.warning(f"{seconds:.2f} s of audio (< 10 s): switching to the configured segment size "
                                "(override_model_segment_size), as the reference does for short files")

        dm = self._demixer()
        eng = dm.engine
        norm = lambda w: eng.normalize(w, self.normalization_threshold, self.amplification_threshold)   # noqa: E731
        mix = norm(np.ascontiguousarray(mix, np.float32))
        source = dm.demix(mix)

        training = self.model_data.get("training", {}) or {}
        if not isinstance(source, dict):
            # single-target model without residual: one array (mdxc_separator.py:213-225)
            files = []
            if self._wanted(self.primary_stem_name):
                if not isinstance(self.primary_source, np.ndarray):
                    self.primary_source = source.T
                self.primary_stem_output_path = self._emit_stem(self.primary_stem_name, self.primary_source, custom_output_names, files)
            return files
        stems = [training["target_instrument"]] if training.get("target_instrument") else list(training.get("instruments") or [])
        if self.process_all_stems and len(stems) > 2:
            files = []
            for name in stems:
                self._emit_stem(name, norm(source[name]).T, custom_output_names, files)
            return files
        if not isinstance(self.primary_source, np.ndarray):
            self.primary_source = norm(source[self.primary_stem_name]).T
        if not isinstance(self.secondary_source, np.ndarray):
            self.secondary_source = norm(source[self.secondary_stem_name]).T
        return self._emit_pair(custom_output_names)
