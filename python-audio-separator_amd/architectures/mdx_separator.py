"""MDXSeparator on the HIP engine: drop-in for audio_separator/separator/architectures/mdx_separator.py.

Same constructor (``common_config``, ``arch_config``), attributes, ``load_model`` / ``separate`` / ``demix`` /
``run_model`` / ``initialize_model_settings`` / ``initialize_mix`` contract, output naming and ``ValueError``s.
``ort.InferenceSession(model_path)`` (:108-133) is replaced by the ONNX reader + ``asx_net_*``; the array work of
``separate`` (:155-182) -- peak, in-place normalise, demix, ``* peak``, ``mix.T - compensate * primary`` -- is one C call
(``asx_separate``) and the writer's normalise / int16 / interleave another (``asx_pcm16``).
"""
from __future__ import annotations

import numpy as np

from ..common_separator import CommonSeparator
from ..mdx import MDXDemixer


class MDXSeparator(CommonSeparator):
    def __init__(self, common_config, arch_config):
        super().__init__(config=common_config)
        self._read_options(arch_config, (("segment_size", None), ("overlap", None), ("batch_size", 1), ("hop_length", None),
                                         ("enable_denoise", None)))
        self.logger.debug(f"MDX arch params: batch_size={self.batch_size}, segment_size={self.segment_size}, overlap={self.overlap}, "
                          f"hop_length={self.hop_length}, enable_denoise={self.enable_denoise}")
        md = self.model_data                                   # the hash-keyed model parameters (separator.py:786-803)
        self.compensate, self.dim_f, self.n_fft = md["compensate"], md["mdx_dim_f_set"], md["mdx_n_fft_scale_set"]
        self.dim_t = 2 ** md["mdx_dim_t_set"]
        self.config_yaml = md.get("config_yaml")
        # engine knob, not a reference option: chunks per device batch (results do not depend on it)
        self._max_batch = int(arch_config.get("asx_max_batch", 0))
        self._common, self._arch = dict(common_config), dict(arch_config)

        self.load_model()

        self.n_bins = self.trim = self.chunk_size = self.gen_size = 0      # filled by initialize_model_settings
        self.stft = None
        self._reset_file_state()

    def load_model(self):
        """mdx_separator.py:108-133.  ``common_config["asx_state_dict"]`` (a ConvTDFNet state_dict, optional, with
        ``asx_net_config``) bypasses the file for callers that hold the weights in memory."""
        common = dict(self._common)
        common["logger"] = self.logger
        self._dm = MDXDemixer(common, self._arch, state_dict=common.get("asx_state_dict"),
                              net_config=common.get("asx_net_config"), max_batch=self._max_batch)
        self.engine = self._dm.engine
        self.model_run = self._dm.engine.net_forward     # spek [B, 4, dim_f, dim_t] -> same (mdx_separator.py:123)

    def initialize_model_settings(self):
        """mdx_separator.py:205-228."""
        self._dm.initialize_model_settings()
        self.n_bins, self.trim = self._dm.n_bins, self._dm.trim
        self.chunk_size, self.gen_size, self.stft = self._dm.chunk_size, self._dm.gen_size, self._dm.stft

    def initialize_mix(self, mix, is_ckpt=False):
        """mdx_separator.py:230-291 (unused by demix in the reference as well): chunk tensor + pad, as numpy."""
        if mix.shape[0] != 2:
            raise ValueError(f"Expected a 2-channel audio signal, but got {mix.shape[0]} channels")
        self.initialize_model_settings()
        n = mix.shape[-1]
        if is_ckpt:
            pad = self.gen_size + self.trim - (n % self.gen_size)
            mixture = np.concatenate((np.zeros((2, self.trim), "float32"), mix, np.zeros((2, pad), "float32"),
                                      np.zeros((2, self.trim), "float32")), 1)
            waves = [mixture[:, i * self.gen_size: i * self.gen_size + self.chunk_size]
                     for i in range(mixture.shape[-1] // self.gen_size)]
        else:
            pad = self.gen_size - n % self.gen_size
            mix_p = np.concatenate((np.zeros((2, self.trim)), mix, np.zeros((2, pad)), np.zeros((2, self.trim))), 1)
            waves, i = [], 0
            while i < n + pad:
                waves.append(np.array(mix_p[:, i: i + self.chunk_size]))
                i += self.gen_size
        return np.asarray(waves, dtype=np.float32), pad

    def demix(self, mix, is_match_mix=False):
        """mdx_separator.py:293-412: float32 [2, N] -> [2, N]."""
        out = self._dm.demix(mix, is_match_mix=is_match_mix)
        self.n_bins, self.trim = self._dm.n_bins, self._dm.trim
        self.chunk_size, self.gen_size, self.stft = self._dm.chunk_size, self._dm.gen_size, self._dm.stft
        return out

    def run_model(self, mix, is_match_mix=False):
        """mdx_separator.py:414-450."""
        return self._dm.run_model(mix, is_match_mix=is_match_mix)

    def _separate_on_device(self, custom_output_names):
        """The same steps with every array in HBM (RIFF/WAVE input at the model's rate): data chunk -> pinned -> device ->
        asx_pcm_decode_dev -> asx_separate_dev -> [host mirrors of the float stems, pinned, for ``primary_source`` /
        ``secondary_source``] -> asx_pcm16_rows_dev per written stem -> int16 back -> container.  The float stems are never
        uploaded again.  Returns None when the file needs the host decoder (the caller continues on the generic path)."""
        mix = self._device_mix(self.audio_file_path)
        if mix is None:
            return None
        import torch
        t0 = self._now()
        self.initialize_model_settings()
        n = mix.shape[1]
        primary = torch.empty((n, 2), dtype=torch.float32, device=mix.device)
        secondary = torch.empty((n, 2), dtype=torch.float32, device=mix.device)
        self.engine.separate_dev(mix.data_ptr(), n, self.normalization_threshold, self.amplification_threshold, self.compensate,
                                 primary.data_ptr(), secondary.data_ptr(), stream=self._stream())
        t0 = self._tick("demix", t0)
        if not isinstance(self.primary_source, np.ndarray):
            self.primary_source = self._host_stem(primary)
        if not isinstance(self.secondary_source, np.ndarray):
            self.secondary_source = self._host_stem(secondary)
        self._sync()                     # the host mirrors are complete before anyone can read primary_source / secondary_source
        self._tick("stems_d2h", t0)
        return self._emit_pair(custom_output_names)

    def separate(self, audio_file_path, custom_output_names=None):
        """mdx_separator.py:135-203."""
        self._begin_file(audio_file_path)
        if not self.invert_using_spec:
            files = self._separate_on_device(custom_output_names)
            if files is not None:
                return files
        mix = self.prepare_mix(self.audio_file_path)
        if mix.shape[0] != 2:
            msg = f"Expected a 2-channel audio signal, but got {mix.shape[0]} channels"
            self.logger.error(msg)
            raise ValueError(msg)
        mix = np.ascontiguousarray(mix, np.float32)
        self.initialize_model_settings()
        need_primary = not isinstance(self.primary_source, np.ndarray)
        need_secondary = not isinstance(self.secondary_source, np.ndarray)
        # peak / normalise(mix) in place / demix * peak / mix.T - compensate * primary (or invert_stem): MDXDemixer.separate_stems
        primary, secondary = self._dm.separate_stems(mix)
        if need_primary:
            self.primary_source = primary
        if need_secondary:
            self.secondary_source = secondary

        return self._emit_pair(custom_output_names)
