"""Host-side mirror of the MDXC plugin's demix path: TFC-TDF v3 (MDX23C), BS-Roformer and Mel-Band Roformer models.

Reference: architectures/mdxc_separator.py (MDXCSeparator.__init__ :22, load_model :76, demix :257 -- Roformer branch
:272-343, TFC branch :345-404, the stem dictionary :406-468), uvr_lib_v5/tfc_tdf_v3.py, uvr_lib_v5/roformer/*.py and the
loader stack under roformer/ (roformer_config.py here).  Array level only; the file-level plugin class on top of this is
architectures/mdxc_separator.py.  No CPU path.
"""
from __future__ import annotations

import logging

import numpy as np

from .engine import Engine, MDXConfig, RofConfig, V3Config
from .mdx import _device_index


def mel_band_layout(sr: int, n_fft: int, n_mels: int):
    """(first bin, bin count) of every band of MelBandRoformer (uvr_lib_v5/roformer/mel_band_roformer.py:279-300): the
    support of librosa.filters.mel(sr=sr, n_fft=n_fft, n_mels=n_mels) -- Slaney mel scale, triangles between consecutive
    mel points -- with bin 0 forced into the first band and the last bin into the last."""
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(float(sr) / 2), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, len(fftfreqs)), dtype=np.float32)
    for i in range(n_mels):
        weights[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, np.newaxis].astype(np.float32)
    weights[0][0] = 1.0
    weights[-1, -1] = 1.0
    pattern = weights > 0
    if not pattern.any(axis=0).all():
        raise ValueError("all frequencies need to be covered by all bands for now")
    starts, counts = [], []
    for i in range(n_mels):
        idx = np.nonzero(pattern[i])[0]
        if len(idx) == 0 or idx[-1] - idx[0] + 1 != len(idx):
            raise NotImplementedError(f"mel band {i} is empty or not one contiguous run of bins")
        starts.append(int(idx[0]))
        counts.append(int(len(idx)))
    return starts, counts


def _get(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def stft_window_table(window_fn, win_length: int, n_fft: int) -> np.ndarray:
    """BSRoformer's ``stft_window_fn`` (bs_roformer.py:333, 386: ``partial(default(stft_window_fn, torch.hann_window), stft_win_length)``)
    as the table torch.stft / istft use: the function evaluated over ``win_length`` in float32 and zero padded to ``n_fft`` at both
    ends.  ``window_fn`` is a callable, or the dotted name a YAML carries (``torch.hamming_window``, ``hamming_window``); only functions of
    the ``torch`` namespace are resolved from names."""
    import torch
    if isinstance(window_fn, str):
        name = window_fn.rsplit(".", 1)[-1]
        if not name.endswith("_window") or not hasattr(torch, name):
            raise NotImplementedError(f"stft_window_fn {window_fn!r}: not a torch window function")
        window_fn = getattr(torch, name)
    w = window_fn(int(win_length)).to(torch.float32).cpu().numpy()
    out = np.zeros(int(n_fft), np.float32)
    off = (int(n_fft) - int(win_length)) // 2
    out[off:off + int(win_length)] = w
    return out


class MDXCDemixer:
    """``common_config["model_data"]`` is the parsed model YAML (separator.py:758-777)."""

    def __init__(self, common_config: dict, arch_config: dict, state_dict: dict | None = None, max_batch: int = 0):
        self.logger = common_config.get("logger") or logging.getLogger(__name__)
        self.torch_device = common_config.get("torch_device")
        self.model_data = common_config.get("model_data") or {}
        self.model_path = common_config.get("model_path")
        self.primary_stem_name = common_config.get("primary_stem_name")
        self.secondary_stem_name = common_config.get("secondary_stem_name")
        self.segment_size = arch_config.get("segment_size", 256)
        self.override_model_segment_size = arch_config.get("override_model_segment_size", False)
        self.overlap = arch_config.get("overlap", 8)
        self.batch_size = arch_config.get("batch_size", 1)
        self.pitch_shift = arch_config.get("pitch_shift", 0)
        if self.pitch_shift != 0:
            from .vr import reference_wav_resolution
            if reference_wav_resolution() != "sinc_fastest":
                # spec_utils.py:33-35: macOS on ARM resamples the pitch round trip with resampy's kaiser_best
                raise NotImplementedError("pitch_shift on a platform where the reference uses resampy (kaiser_best)")
        audio, model, training = (self.model_data.get(k, {}) for k in ("audio", "model", "training"))
        # CommonSeparator._detect_roformer_model (common_separator.py:521-543): the flag, or "roformer" in the path / name
        self.is_roformer = bool(self.model_data.get("is_roformer")) or any(
            t and "roformer" in str(t).lower() for t in (self.model_path, common_config.get("model_name")))
        self.instruments = list(training.get("instruments") or [])
        self.target_instrument = training.get("target_instrument")
        self.is_primary_stem_main_target = bool(self.target_instrument)
        if self.primary_stem_name is None:
            self.primary_stem_name = self.target_instrument or (self.instruments[0] if self.instruments else "Primary")
        seg = self.segment_size if self.override_model_segment_size else _get(self.model_data, "inference", "dim_t")
        self.mdx_segment_size = int(seg)
        self.sample_rate = audio.get("sample_rate", common_config.get("sample_rate", 44100))
        if self.is_roformer:
            self._init_roformer(audio, model, state_dict, max_batch)
            return
        scale = list(model.get("scale", [2, 2]))
        if scale != [2, 2]:
            raise NotImplementedError(f"scale {scale} (only [2, 2] is supported)")
        self.v3 = V3Config(num_channels=audio.get("num_channels", 2), num_subbands=model["num_subbands"],
                           num_scales=model["num_scales"], num_blocks_per_scale=model["num_blocks_per_scale"],
                           num_channels_model=model["num_channels"], growth=model["growth"],
                           bottleneck_factor=model["bottleneck_factor"], norm=model.get("norm"),
                           act=model.get("act", "gelu"),
                           num_targets=1 if self.target_instrument else len(self.instruments))
        self.engine = Engine(MDXConfig(n_fft=audio["n_fft"], hop_length=audio["hop_length"], dim_f=audio["dim_f"],
                                       segment_size=self.mdx_segment_size, overlap=0.0, max_batch=max_batch),
                             device=_device_index(self.torch_device))
        if state_dict is not None or self.model_path:
            self.load_model(state_dict)

    def _init_roformer(self, audio, model, state_dict, max_batch):
        """BS-Roformer (freqs_per_bands) and Mel-Band Roformer (num_bands) models.  The constructor arguments are the ones
        the reference's loader would pass (roformer_loader.py:123-195 after configuration_normalizer.py), quirks included."""
        from .roformer_config import RoformerLoader
        self.roformer_loader = getattr(self, "roformer_loader", None) or RoformerLoader()
        res = self.roformer_loader.load_model(self.model_path or "", self.model_data, state_dict=state_dict)
        if not res.success:
            raise RuntimeError(res.error_message)
        a = res.args
        self._rof_state = res.state_dict
        n_fft = a["stft_n_fft"]
        win_length = int(a["stft_win_length"])
        if not 0 < win_length <= n_fft:
            raise ValueError(f"stft_win_length {win_length} must be in (0, stft_n_fft = {n_fft}]")
        # the chunk loop derives its hop from the raw YAML (mdxc_separator.py:289-296), the network from the normalised one
        hop = model.get("stft_hop_length") or audio["hop_length"]
        if int(hop) != int(a["stft_hop_length"]):
            raise NotImplementedError(f"chunk hop {hop} differs from the model's stft_hop_length {a['stft_hop_length']}")
        mel = res.model_type == "mel_band_roformer"
        if mel:
            starts, counts = mel_band_layout(a["sample_rate"], n_fft, a["num_bands"])
        else:
            starts, counts = (), tuple(a["freqs_per_bands"])
        self.rof = RofConfig(dim=a["dim"], depth=a["depth"], heads=a["heads"], dim_head=a["dim_head"], num_stems=a["num_stems"],
                             time_transformer_depth=a["time_transformer_depth"], freq_transformer_depth=a["freq_transformer_depth"],
                             mlp_expansion_factor=a["mlp_expansion_factor"], mask_estimator_depth=a["mask_estimator_depth"],
                             freqs_per_bands=tuple(counts), n_out=max(1, len(self.instruments)), mel=mel,
                             band_starts=tuple(starts), stft_normalized=bool(a["stft_normalized"]))
        if not a["stereo"]:
            # (the reference cannot run them either: prepare_mix always yields [2, N] and BSRoformer.forward asserts one channel
            # for a mono model, bs_roformer.py:443-445)
            raise NotImplementedError("mono Roformer models")
        self.engine = Engine(MDXConfig(n_fft=n_fft, hop_length=int(hop), dim_f=n_fft // 2 + 1,
                                       segment_size=self.mdx_segment_size, overlap=0.0, max_batch=max_batch,
                                       win_length=0 if win_length == n_fft else win_length),
                             device=_device_index(self.torch_device))
        window_fn = a.get("stft_window_fn")
        if window_fn is not None:
            self.engine.set_stft_window(stft_window_table(window_fn, win_length, n_fft))
        if self._rof_state is not None:
            self.load_model(self._rof_state)

    def load_model(self, state_dict: dict | None = None):
        """torch.load(model_path) + load_state_dict (mdxc_separator.py:107-110, roformer_loader.py:97-104)."""
        if state_dict is None:
            from .model_files import read_state_dict
            from .roformer_config import read_checkpoint
            state_dict = read_checkpoint(self.model_path) if self.is_roformer else read_state_dict(self.model_path)
        if self.is_roformer:
            self.engine.load_rof(self.rof, state_dict)
        else:
            self.engine.load_v3(self.v3, state_dict)

    def demix(self, mix: np.ndarray):
        """mdxc_separator.py:257-468 for TFC-TDF models: dict of stems, or the primary array."""
        mix = np.asarray(mix, dtype=np.float32)
        if mix.ndim != 2 or mix.shape[0] != 2:
            raise ValueError(f"Expected a 2-channel audio signal, but got {mix.shape[0] if mix.ndim else 0} channels")
        if mix.shape[1] == 0:
            raise ValueError("Audio file is empty or not valid")
        if self.pitch_shift != 0:
            return self._demix_pitched(mix)
        if self.is_roformer:
            return self._demix_roformer(mix)
        out = self.engine.mdxc_demix(mix, int(self.overlap))
        if self.v3.num_targets > 1:
            return {k: out[i] for i, k in enumerate(self.instruments)}
        primary = out[0]
        if self.is_primary_stem_main_target:
            return {self.primary_stem_name: primary, self.secondary_stem_name: mix - primary}
        return primary

    # ---- pitch_shift (mdxc_separator.py:230-243, 268-270, 417-419, 450-466) -----------------------------------------------------
    def change_pitch_semitones(self, y: np.ndarray, sr, semitone_shift):
        """spec_utils.change_pitch_semitones (:783-790): every channel through librosa.resample(res_type="sinc_fastest") to
        sr * 2^(shift / 12) -- on the device (asx_resample_sinc, one mono libsamplerate call per channel)."""
        factor = 2 ** (semitone_shift / 12)
        target = sr * factor
        return self.engine.resample_sinc(np.ascontiguousarray(y, np.float32), float(target) / sr, mono_calls=True), target

    def pitch_fix(self, source, sr_pitched, orig_mix):
        source = self.change_pitch_semitones(source, sr_pitched, self.pitch_shift)[0]
        n = orig_mix.shape[1]                                    # spec_utils.match_array_shapes (:752-769)
        if source.shape[1] > n:
            return source[:, :n]
        if source.shape[1] < n:
            return np.pad(source, ((0, 0), (0, n - source.shape[1])), "constant", constant_values=0)
        return source

    def _demix_pitched(self, orig_mix: np.ndarray):
        """demix with pitch_shift != 0: the mix is resampled to sr * 2^(-p / 12), separated, and every stem resampled back by
        2^(p / 12) and padded / trimmed to the mix's length; the residual stem is taken against the ORIGINAL mix."""
        mix, sr_pitched = self.change_pitch_semitones(orig_mix, self.sample_rate, -self.pitch_shift)
        if self.is_roformer:
            chunk_size = self.engine.cfg.hop_length * (self.mdx_segment_size - 1)
            desired_step = int(self.overlap * self.sample_rate)
            step = chunk_size if desired_step <= 0 else min(desired_step, chunk_size)
            out = self.engine.rof_demix(mix, step)
            num_stems = 1 if self.target_instrument else len(self.instruments)
        else:
            out = self.engine.mdxc_demix(mix, int(self.overlap))
            num_stems = self.v3.num_targets
        if num_stems > 1:
            return {k: self.pitch_fix(out[i], sr_pitched, orig_mix) for i, k in enumerate(self.instruments)}
        primary = self.pitch_fix(out[0], sr_pitched, orig_mix)
        if self.is_primary_stem_main_target:
            return {self.primary_stem_name: primary, self.secondary_stem_name: orig_mix - primary}
        return primary

    def demix_dev(self, mix_d):
        """``demix`` with the (already normalised) mix [2, N] and the stems in HBM: returns (names, CUDA tensor [S, 2, N]) -- the
        stems in the order of ``names``; a single-target model without residual gives ([None], [1, 2, N]).  The residual stem
        ``mix - primary`` (mdxc_separator.py:406-468) is asx_residual_dev.  Only enqueues work."""
        import torch
        n = mix_d.shape[1]
        st = torch.cuda.current_stream(mix_d.device).cuda_stream
        if self.is_roformer:
            chunk_size = self.engine.cfg.hop_length * (self.mdx_segment_size - 1)
            desired_step = int(self.overlap * self.sample_rate)
            step = chunk_size if desired_step <= 0 else min(desired_step, chunk_size)
            n_out = int(self.engine.rof_cfg.n_out)
            multi = not self.target_instrument and len(self.instruments) > 1
            raw = torch.empty((n_out + 1, 2, n), dtype=torch.float32, device=mix_d.device)      # + one row for a residual stem
            self.engine.rof_demix_dev(mix_d.data_ptr(), n, step, raw.data_ptr(), stream=st)
        else:
            n_out = self.v3.num_targets
            multi = n_out > 1
            raw = torch.empty((n_out + 1, 2, n), dtype=torch.float32, device=mix_d.device)
            self.engine.mdxc_demix_dev(mix_d.data_ptr(), n, int(self.overlap), raw.data_ptr(), stream=st)
        if multi:
            return list(self.instruments)[:n_out], raw[:n_out]
        if self.is_primary_stem_main_target:
            self.engine.residual_dev(mix_d.data_ptr(), raw[0].data_ptr(), raw[1].data_ptr(), 2 * n, stream=st)
            return [self.primary_stem_name, self.secondary_stem_name], raw[:2]
        return [None], raw[:1]

    def _demix_roformer(self, mix: np.ndarray):
        """mdxc_separator.py:272-343 + :406-468 for Roformer models."""
        chunk_size = self.engine.cfg.hop_length * (self.mdx_segment_size - 1)
        desired_step = int(self.overlap * self.sample_rate)
        step = chunk_size if desired_step <= 0 else min(desired_step, chunk_size)
        out = self.engine.rof_demix(mix, step)                       # [len(instruments), 2, N]
        num_stems = 1 if self.target_instrument else len(self.instruments)
        if num_stems > 1:
            return {k: out[i] for i, k in enumerate(self.instruments)}
        primary = out[0]
        if self.is_primary_stem_main_target:
            return {self.primary_stem_name: primary, self.secondary_stem_name: mix - primary}
        return primary
