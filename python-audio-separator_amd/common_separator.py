"""The plugin-surface base class: what ``Separator.load_model`` / ``_separate_file`` expect of a model instance.

Mirror of audio_separator/separator/common_separator.py:CommonSeparator (reference file:line cited per member):
same constructor key set (``common_config`` built at separator.py:867-886), same public attributes the orchestrator
reads and writes (``output_dir`` separator.py:1087-1088, ``is_roformer_model`` / ``get_roformer_loading_stats`` :926-929,
``write_audio`` :1379), same stem naming, same file naming and the same per-file state reset.  The sample
arithmetic of the edges (``spec_utils.normalize``, int16 quantisation, channel interleave) runs in libasx.so through
``self.engine``; decode / encode containers stay on the host (audio_io.py).  There is no CPU path for the arithmetic:
a subclass without an engine cannot write audio.
"""
from __future__ import annotations

import functools
import gc
import logging
import os
import re
import threading

import numpy as np

from . import audio_io


class CommonSeparator:
    # Stem vocabulary (common_separator.py:19-53).  The names are part of the contract: the orchestrator, the ensembler presets
    # and the output file names refer to them.  Kept as one table; the class attributes are generated from it below.
    _STEM_LABELS = {
        "ALL_STEMS": "All Stems", "PRIMARY_STEM": "Primary Stem", "SECONDARY_STEM": "Secondary Stem",
        # instruments, and the name of "everything but" each of them
        "VOCAL_STEM": "Vocals", "INST_STEM": "Instrumental",
        "OTHER_STEM": "Other", "NO_OTHER_STEM": "No Other",
        "BASS_STEM": "Bass", "NO_BASS_STEM": "No Bass",
        "DRUM_STEM": "Drums", "NO_DRUM_STEM": "No Drums",
        "GUITAR_STEM": "Guitar", "NO_GUITAR_STEM": "No Guitar",
        "PIANO_STEM": "Piano", "NO_PIANO_STEM": "No Piano",
        "SYNTH_STEM": "Synthesizer", "NO_SYNTH_STEM": "No Synthesizer",
        "STRINGS_STEM": "Strings", "NO_STRINGS_STEM": "No Strings",
        "WOODWINDS_STEM": "Woodwinds", "NO_WOODWINDS_STEM": "No Woodwinds",
        "BRASS_STEM": "Brass", "NO_BRASS_STEM": "No Brass",
        "WIND_INST_STEM": "Wind Inst", "NO_WIND_INST_STEM": "No Wind Inst",
        "NO_STEM": "No ",
        # karaoke / backing-vocal models
        "LEAD_VOCAL_STEM": "lead_only", "BV_VOCAL_STEM": "backing_only",
        "LEAD_VOCAL_STEM_I": "with_lead_vocals", "BV_VOCAL_STEM_I": "with_backing_vocals",
        "LEAD_VOCAL_STEM_LABEL": "Lead Vocals", "BV_VOCAL_STEM_LABEL": "Backing Vocals",
    }
    locals().update(_STEM_LABELS)

    _CONFIG_KEYS = ("log_level", "torch_device", "torch_device_cpu", "torch_device_mps", "onnx_execution_provider",
                    "model_name", "model_path", "model_data", "output_dir", "output_format", "output_bitrate",
                    "normalization_threshold", "amplification_threshold", "enable_denoise", "output_single_stem",
                    "invert_using_spec", "sample_rate", "use_soundfile")

    def __init_subclass__(cls, **kw):
        """Every architecture's ``separate`` returns only after the container writes it started have finished (the built-in WAV
        writer runs on worker threads so that the files of a song are written concurrently and overlap the next stem's int16
        pass; the orchestrator may open the files as soon as ``separate`` returns)."""
        super().__init_subclass__(**kw)
        inner = cls.__dict__.get("separate")
        if inner is not None:
            @functools.wraps(inner)
            def separate(self, *args, **kwargs):
                self._in_separate = True
                try:
                    result = inner(self, *args, **kwargs)
                except BaseException:
                    self._in_separate = False
                    self._drain_writes(raise_errors=False)      # the caller's exception wins over a writer's
                    raise
                self._in_separate = False
                self._drain_writes()
                return result
            cls.separate = separate

    def __init__(self, config: dict):
        """common_separator.py:55-148."""
        self.logger = config.get("logger") or logging.getLogger("audio_separator_amd")
        for key in self._CONFIG_KEYS:
            setattr(self, key, config.get(key))
        if self.model_data is None:
            self.model_data = {}
        self.engine = None                       # the libasx.so handle; set by the architecture subclass
        self.asx_profile_file = bool(config.get("asx_profile_file"))   # per-phase wall times of separate() in file_timings
        self.file_timings = {}
        self._pending_writes = []                # (thread, error box) of container writes in flight

        self.roformer_loader = None
        self.is_roformer_model = self._detect_roformer_model()
        if self.is_roformer_model:
            self._initialize_roformer_loader()

        self.primary_stem_name = None
        self.secondary_stem_name = None
        self.input_bit_depth = None
        self.input_subtype = None

        training = self.model_data.get("training") if isinstance(self.model_data, dict) else None
        instruments = (training or {}).get("instruments") if isinstance(training, dict) else None
        if instruments:
            target = training.get("target_instrument")
            # a target that is listed second names the model's actual output (common_separator.py:112-122)
            if target and len(instruments) >= 2 and instruments[0] != target and instruments[1] == target:
                self.primary_stem_name, self.secondary_stem_name = instruments[1], instruments[0]
            else:
                self.primary_stem_name = instruments[0]
                self.secondary_stem_name = instruments[1] if len(instruments) > 1 else self.secondary_stem(instruments[0])
        if self.primary_stem_name is None:
            self.primary_stem_name = self.model_data.get("primary_stem", "Vocals")
            self.secondary_stem_name = self.secondary_stem(self.primary_stem_name)

        self.is_karaoke = self.model_data.get("is_karaoke", False)
        self.is_bv_model = self.model_data.get("is_bv_model", False)
        self.bv_model_rebalance = self.model_data.get("is_bv_model_rebalanced", 0)

        self.logger.debug(f"Common params: model_name={self.model_name}, model_path={self.model_path}, "
                          f"output_dir={self.output_dir}, output_format={self.output_format}")
        self.logger.debug(f"Common params: primary_stem_name={self.primary_stem_name}, "
                          f"secondary_stem_name={self.secondary_stem_name}")

        self._reset_file_state()
        self.cached_sources_map = {}

    # what one input file leaves behind (common_separator.py:136-147, 509-520): reset by clear_file_specific_paths
    _FILE_STATE = ("audio_file_path", "audio_file_base", "primary_source", "secondary_source", "primary_stem_output_path",
                   "secondary_stem_output_path")

    def _reset_file_state(self):
        for name in self._FILE_STATE:
            setattr(self, name, None)
        self._dev_stems = {}          # id(host array) -> (host array, device tensor [N, 2]): stems that never left HBM
        self._file_seconds = None     # duration of the current input, probed once per file

    # ---- steps every architecture's separate() shares ---------------------------------------------------------------
    def _read_options(self, arch_config: dict, table):
        """``table`` = ((key, default), ...): the architecture options the orchestrator passes (separator.py:125-128)."""
        for key, default in table:
            setattr(self, key, arch_config.get(key, default))

    def _begin_file(self, audio_file_path: str):
        self.audio_file_path = audio_file_path
        self.audio_file_base = os.path.splitext(os.path.basename(audio_file_path))[0]
        self._dev_stems = {}
        self._file_seconds = None
        self.file_timings = {}        # seconds per phase of this file when ``asx_profile_file`` is set (bench.py file_level)

    # ---- device-resident file path (load / normalise / write edges without host round trips) -----------------
    # ---- concurrent container writes -----------------------------------------------------------------------------
    def _write_wav_async(self, stem_path, pcm, subtype):
        """audio_io.write_wav on a worker thread (the file write releases the GIL); drained before ``separate`` returns."""
        # a write_audio call from outside separate() (the orchestrator's ensemble output, separator.py:1379) is synchronous:
        # nobody would drain it
        if os.environ.get("ASX_ASYNC_WRITES", "1") == "0" or not getattr(self, "_in_separate", False):
            audio_io.write_wav(stem_path, pcm, self.sample_rate, subtype)
            return
        box = []

        def work():
            try:
                audio_io.write_wav(stem_path, pcm, self.sample_rate, subtype)
            except BaseException as e:          # re-raised by _drain_writes on the caller's thread
                box.append(e)
        t = threading.Thread(target=work, name="asx-wav-writer", daemon=True)
        t.start()
        self._pending_writes.append((t, box))

    def _drain_writes(self, raise_errors: bool = True):
        t0 = self._now()
        pending, self._pending_writes = self._pending_writes, []
        err = None
        for t, box in pending:
            t.join()
            if box and err is None:
                err = box[0]
        if pending and getattr(self, "asx_profile_file", False):
            self.file_timings["write_drain"] = self.file_timings.get("write_drain", 0.0) + (self._now() - t0)
        if err is not None:
            if raise_errors:
                raise err
            self.logger.error(f"a container write failed: {err}")

    def _tick(self, phase: str, t0: float) -> float:
        """Accumulate wall time of ``phase`` since ``t0`` (device drained first) when per-file profiling is on."""
        if not getattr(self, "asx_profile_file", False):
            return t0
        import time
        self._sync()
        t1 = time.perf_counter()
        self.file_timings[phase] = self.file_timings.get(phase, 0.0) + (t1 - t0)
        return t1

    def _now(self) -> float:
        import time
        return time.perf_counter()

    def _torch_device(self):
        import torch
        return torch.device("cuda", self.engine.device)

    def _stream(self) -> int:
        import torch
        return torch.cuda.current_stream(self._torch_device()).cuda_stream

    def _sync(self):
        import torch
        if self.engine is not None and torch.cuda.is_available():
            torch.cuda.current_stream(self._torch_device()).synchronize()

    def _fast_file_path_enabled(self) -> bool:
        if self.engine is None or os.environ.get("ASX_FILE_FASTPATH", "1") == "0" or not hasattr(self.engine, "pcm_decode_dev"):
            return False
        try:
            import torch
            return torch.cuda.is_available()      # torch is the allocator of the pinned / device staging buffers
        except Exception:
            return False

    def _device_mix(self, path, check_silent=True):
        """prepare_mix for a RIFF/WAVE file at the model's rate WITHOUT a host float array: the data chunk is read into pinned
        memory, copied to HBM once and converted to the planar float32 mix [2, N] on the device (asx_pcm_decode_dev: the same
        x / 2^(bits-1) conversion libsndfile applies under librosa.load, common_separator.py:252).  Returns a CUDA tensor, or
        None when the file needs the host decoder (other container, other rate, exotic subtype) -- the caller then takes
        prepare_mix.  Raises the reference's ValueError for a silent file (:268-271) unless ``check_silent`` is False: the VR
        plugin never calls prepare_mix (vr_separator.py:255-291 loads with librosa directly), so a silent file must come out the
        same whichever decode path it took (the writer's "nothing to write" branch handles silent stems)."""
        if not self._fast_file_path_enabled() or not isinstance(path, str):
            return None
        import torch
        try:
            info = audio_io.wav_info(path)
        except (audio_io.AudioIOError, OSError):
            return None
        if info["samplerate"] != self.sample_rate or info["subtype"] not in self.engine.PCM_FORMATS or info["frames"] < 1 or info["channels"] > 2:
            return None                # (more than two channels: prepare_mix + the reference's "Expected a 2-channel" error)
        t0 = self._now()
        # what _probe_bit_depth records for the writer (common_separator.py:231-250)
        self.input_subtype = st = info["subtype"]
        self.input_bit_depth = 16 if st == "PCM_16" else (24 if st == "PCM_24" else 32)
        self._file_seconds = info["frames"] / float(info["samplerate"])
        frames, ch = info["frames"], info["channels"]
        nbytes = frames * ch * info["bits"] // 8
        staged = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        with open(path, "rb") as f:
            f.seek(info["data_offset"])
            got = f.readinto(memoryview(staged.numpy()))
        if got != nbytes:
            return None
        t0 = self._tick("read", t0)
        dev = self._torch_device()
        raw = staged.to(dev, non_blocking=True)
        mix = torch.empty((2, frames), dtype=torch.float32, device=dev)
        peak = self.engine.pcm_decode_dev(raw.data_ptr(), frames, ch, st, mix.data_ptr(), stream=self._stream())
        self._tick("h2d_decode", t0)
        if check_silent and not peak > 0.0:
            msg = f"Audio file {path} is empty or not valid"
            self.logger.error(msg)
            raise ValueError(msg)
        return mix

    def _host_stem(self, dev_stem):
        """A device stem [N, 2] as the numpy array the reference leaves in ``primary_source`` / ``secondary_source`` (pinned
        staging, one asynchronous copy) and remember which device tensor it mirrors, so write_audio can quantise on the device
        instead of uploading the array again."""
        import torch
        host = torch.empty(dev_stem.shape, dtype=dev_stem.dtype, pin_memory=True)
        host.copy_(dev_stem, non_blocking=True)
        arr = host.numpy()
        # the mirror is READ-ONLY: write_audio quantises from the device tensor, so an in-place edit of the host array between
        # separate() and write_audio would be silently dropped -- it raises instead (assign a NEW array to take the generic path)
        arr.flags.writeable = False
        self._dev_stems[id(arr)] = (arr, dev_stem, host, "rows")
        return arr

    def _host_planar_stems(self, dev_stems):
        """Device stems [S, 2, N] -> their pinned host mirror [S, 2, N] plus the [N, 2] VIEWS (``source[i].T``) the reference hands to
        write_audio, each registered against its planar device tensor (the Demucs / MDXC layouts)."""
        import torch
        host = torch.empty(dev_stems.shape, dtype=dev_stems.dtype, pin_memory=True)
        host.copy_(dev_stems, non_blocking=True)
        arr = host.numpy()
        arr.flags.writeable = False        # as in _host_stem: the views below inherit it
        views = []
        for i in range(arr.shape[0]):
            v = arr[i].T
            self._dev_stems[id(v)] = (v, dev_stems[i], host, "planar")
            views.append(v)
        return arr, views

    def _device_stem_entry(self, stem_source):
        hit = self._dev_stems.get(id(stem_source))
        return hit if hit is not None and hit[0] is stem_source else None

    def _device_stem_for(self, stem_source):
        hit = self._device_stem_entry(stem_source)
        return hit[1] if hit is not None else None

    def _wanted(self, stem_name: str) -> bool:
        """``output_single_stem`` filter (mdx_separator.py:185,193)."""
        return not self.output_single_stem or self.output_single_stem.lower() == stem_name.lower()

    def _emit_stem(self, stem_name: str, source, custom_output_names, files: list) -> str:
        path = self.get_stem_output_path(stem_name, custom_output_names)
        self.logger.info(f"{stem_name} -> {path}")
        self.final_process(path, source, stem_name)
        files.append(path)
        return path

    def _emit_pair(self, custom_output_names) -> list:
        """Secondary stem first, then the primary one -- the order the reference returns them in (mdx_separator.py:184-203)."""
        files = []
        if self._wanted(self.secondary_stem_name):
            self.secondary_stem_output_path = self._emit_stem(self.secondary_stem_name, self.secondary_source, custom_output_names, files)
        if self._wanted(self.primary_stem_name):
            self.primary_stem_output_path = self._emit_stem(self.primary_stem_name, self.primary_source, custom_output_names, files)
        return files

    # ---- stem naming -------------------------------------------------------
    def secondary_stem(self, primary_stem: str):
        """common_separator.py:150-159."""
        primary_stem = primary_stem if primary_stem else self.NO_STEM
        if primary_stem in self.STEM_PAIR_MAPPER:
            return self.STEM_PAIR_MAPPER[primary_stem]
        if self.NO_STEM in primary_stem:
            return primary_stem.replace(self.NO_STEM, "")
        return f"{self.NO_STEM}{primary_stem}"

    def separate(self, audio_file_path, custom_output_names=None):
        raise NotImplementedError("This method should be overridden by subclasses.")

    def final_process(self, stem_path, source, stem_name):
        """common_separator.py:167-174."""
        self.logger.debug(f"{stem_name}: writing {stem_path}")
        self.write_audio(stem_path, source)
        return {stem_name: source}

    # ---- source cache (common_separator.py:176-215) -----------------------
    def cached_sources_clear(self):
        self.cached_sources_map = {}

    def cached_source_callback(self, model_architecture, model_name=None):
        model, sources = None, None
        for key, value in self.cached_sources_map[model_architecture].items():
            if model_name in key:
                model, sources = key, value
        return model, sources

    def cached_model_source_holder(self, model_architecture, sources, model_name=None):
        self.cached_sources_map[model_architecture] = {**self.cached_sources_map.get(model_architecture, {}),
                                                       **{model_name: sources}}

    # ---- decode edge -------------------------------------------------------
    def _probe_bit_depth(self, path):
        """common_separator.py:231-250: remember the input's sample format so the writer can keep it."""
        try:
            self.input_subtype = audio_io.info(path)["subtype"]
            st = self.input_subtype
            if "PCM_16" in st or st == "PCM_S8":
                self.input_bit_depth = 16
            elif "PCM_24" in st:
                self.input_bit_depth = 24
            elif "PCM_32" in st or "FLOAT" in st or "DOUBLE" in st:
                self.input_bit_depth = 32
            else:
                self.input_bit_depth = 16
                self.logger.warning(f"Unknown audio subtype {st}, defaulting to 16-bit output")
            self.logger.info(f"Input audio subtype: {st}, bit depth: {self.input_bit_depth}")
        except Exception as e:   # the reference swallows every failure here too
            self.logger.warning(f"no container info for {path} ({e}): stems will be written as 16-bit PCM")
            self.input_bit_depth, self.input_subtype = 16, "PCM_16"

    def prepare_mix(self, mix):
        """common_separator.py:217-282: path -> float32 [2, N] at ``sample_rate``; an ndarray [N, ch] is transposed;
        mono is duplicated; an all-zero file raises ValueError."""
        audio_path = mix
        if not isinstance(mix, np.ndarray):
            self._probe_bit_depth(mix)
            mix, sr = audio_io.load(mix, mono=False, sr=self.sample_rate)
            self.logger.debug(f"decoded {audio_path}: {mix.shape[-1]} samples x {mix.shape[0] if mix.ndim > 1 else 1} channel(s) at {sr} Hz")
        else:
            if self.input_bit_depth is None:
                self.input_bit_depth, self.input_subtype = 16, "PCM_16"
            mix = mix.T
        if isinstance(audio_path, str) and not np.any(mix):
            msg = f"Audio file {audio_path} is empty or not valid"
            self.logger.error(msg)
            raise ValueError(msg)
        if mix.ndim == 1:
            mix = np.asfortranarray([mix, mix])
        return mix

    # ---- encode edge -------------------------------------------------------
    def _require_engine(self):
        if self.engine is None:
            raise RuntimeError("no libasx.so engine bound to this separator: the writer's normalise / quantise runs on the "
                               "GPU and has no CPU path")
        return self.engine

    def write_audio(self, stem_path: str, stem_source):
        """common_separator.py:284-303."""
        if self.audio_file_path:
            if self._file_seconds is None:           # probed once per input, not once per stem
                self._file_seconds = audio_io.duration(self.audio_file_path)
            secs = self._file_seconds
            self.logger.info(f"Audio duration is {secs / 3600:.2f} hours ({secs:.2f} seconds).")
        if self.use_soundfile:
            self.write_audio_soundfile(stem_path, stem_source)
        else:
            self.write_audio_pydub(stem_path, stem_source)

    def _stereo_rows(self, stem_source):
        a = np.asarray(stem_source)
        if a.ndim != 2 or a.shape[1] != 2:
            raise ValueError(f"write_audio expects a [N, 2] stem, got {a.shape}")
        return a

    def write_audio_pydub(self, stem_path: str, stem_source):
        """common_separator.py:305-397.  normalize -> silence check -> (x * 32767).astype(int16) -> interleave is one
        device pass (asx_pcm16, bit-exact); the container is written by pydub when it is installed, else by the
        built-in WAV writer (int16 widened exactly like ffmpeg's s16 -> s32 / pcm_s24le conversion)."""
        eng = self._require_engine()
        a = self._stereo_rows(stem_source)
        t0 = self._now()
        entry = self._device_stem_entry(stem_source)
        dev_stem = entry[1] if entry is not None else None
        if dev_stem is not None:
            # the stem is still in HBM (same array object separate() produced): normalise + quantise there, bring back int16 only
            import torch
            planar = entry[3] == "planar"                     # [2, N] (Demucs / MDXC stems) or [N, 2] (asx_separate_dev)
            n = dev_stem.shape[1] if planar else dev_stem.shape[0]
            pcm_dev = torch.empty((n, 2), dtype=torch.int16, device=dev_stem.device)
            quantise = eng.pcm16_planar_dev if planar else eng.pcm16_rows_dev
            peak = quantise(dev_stem.data_ptr(), n, self.normalization_threshold, self.amplification_threshold,
                            pcm_dev.data_ptr(), stream=self._stream())
            pcm_host = torch.empty((n, 2), dtype=torch.int16, pin_memory=True)
            pcm_host.copy_(pcm_dev, non_blocking=True)
            self._sync()
            pcm = pcm_host.numpy()
            # written: the device stem and its pinned mirror are released with this entry (planar stems share one mirror, which
            # lives until the last of them is written)
            self._dev_stems.pop(id(stem_source), None)
        elif a.dtype == np.int16:
            pcm, peak = np.ascontiguousarray(a), float(np.abs(a).max()) if a.size else 0.0
        else:
            if a.shape[0] == 0:
                self.logger.warning(f"{stem_path}: nothing to write (the stem is empty or silent)")
                return
            pcm, peak = eng.pcm16(a, self.normalization_threshold, self.amplification_threshold)
        t0 = self._tick("pcm16_d2h", t0)
        if peak < 1e-6:
            self.logger.warning(f"{stem_path}: nothing to write (the stem is empty or silent)")
            return
        if self.output_dir:
            os.makedirs(self.output_dir, exist_ok=True)
            stem_path = os.path.join(self.output_dir, stem_path)
        depth = self.input_bit_depth if self.input_bit_depth is not None else 16
        self.logger.info(f"{stem_path}: {depth}-bit output (the input's bit depth)")
        file_format = stem_path.lower().split(".")[-1]
        pydub = audio_io._optional("pydub")
        if pydub is not None and hasattr(pydub, "AudioSegment") and hasattr(pydub.AudioSegment, "export"):
            seg = pydub.AudioSegment(pcm.reshape(-1).tobytes(), frame_rate=self.sample_rate, sample_width=2, channels=2)
            fmt = {"m4a": "mp4", "mka": "matroska"}.get(file_format, file_format)
            params = {"format": fmt}
            bitrate = "320k" if fmt == "mp3" and self.output_bitrate is None else self.output_bitrate
            if bitrate:
                params["bitrate"] = bitrate
            if fmt in ("wav", "flac"):
                if depth == 16:
                    params["parameters"] = ["-sample_fmt", "s16"]
                else:
                    params["parameters"] = ["-sample_fmt", "s32"]
                    if fmt == "wav":
                        params["codec"] = "pcm_s24le" if depth == 24 else "pcm_s32le"
            try:
                seg.export(stem_path, **params)
            except (IOError, ValueError) as e:
                self.logger.error(f"{stem_path}: the container write failed: {e}")
            return
        if file_format != "wav":
            raise audio_io.AudioIOError(f"writing .{file_format} needs pydub + ffmpeg (not installed); WAV is built in")
        self._write_wav_async(stem_path, pcm, {16: "PCM_16", 24: "PCM_24", 32: "PCM_32"}.get(depth, "PCM_16"))
        self._tick("container_write", t0)

    def write_audio_soundfile(self, stem_path: str, stem_source):
        """common_separator.py:399-461: normalise (device), keep the input's subtype, hand floats to the container."""
        eng = self._require_engine()
        a = self._stereo_rows(stem_source)
        if a.shape[0] == 0:
            self.logger.warning(f"{stem_path}: nothing to write (the stem is empty or silent)")
            return
        buf = eng.normalize(np.ascontiguousarray(a, np.float32), self.normalization_threshold, self.amplification_threshold)
        if np.max(np.abs(buf)) < 1e-6:
            self.logger.warning(f"{stem_path}: nothing to write (the stem is empty or silent)")
            return
        if self.output_dir:
            os.makedirs(self.output_dir, exist_ok=True)
            stem_path = os.path.join(self.output_dir, stem_path)
        if self.input_subtype:
            subtype = self.input_subtype
        elif self.input_bit_depth:
            subtype = {16: "PCM_16", 24: "PCM_24", 32: "PCM_32"}.get(self.input_bit_depth, "PCM_16")
        else:
            subtype = "PCM_16"
        sf = audio_io._optional("soundfile")
        try:
            if sf is not None and hasattr(sf, "write"):
                sf.write(stem_path, buf, self.sample_rate, subtype=subtype)
            else:
                audio_io.write_wav(stem_path, buf, self.sample_rate, subtype)
        except Exception as e:
            self.logger.error(f"{stem_path}: the container write failed: {e}")

    # ---- per-file state ------------------------------------------------------
    def clear_gpu_cache(self):
        """common_separator.py:463-474.  The engine's workspaces are sized once and reused across files (they are the
        point of keeping 288 GB resident); only Python garbage and torch's caching allocator are trimmed."""
        self._dev_stems = {}
        gc.collect()
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.empty_cache()
        except Exception:
            pass

    def clear_file_specific_paths(self):
        """common_separator.py:476-489."""
        self.logger.info("per-file state reset (paths, sources, stems)")
        self._reset_file_state()

    # ---- file naming -------------------------------------------------------
    def sanitize_filename(self, filename):
        """common_separator.py:491-498."""
        s = re.sub(r'[<>:"/\\|?*]', "_", filename)
        s = re.sub(r"_+", "_", s)
        return s.strip("_. ")

    def get_stem_output_path(self, stem_name, custom_output_names):
        """common_separator.py:500-519: names are relative to ``output_dir``."""
        ext = self.output_format.lower()
        if custom_output_names:
            lowered = {k.lower(): v for k, v in custom_output_names.items()}
            if stem_name.lower() in lowered:
                return os.path.join(f"{self.sanitize_filename(lowered[stem_name.lower()])}.{ext}")
        return os.path.join(f"{self.sanitize_filename(self.audio_file_base)}_({self.sanitize_filename(stem_name)})_"
                            f"{self.sanitize_filename(self.model_name)}.{ext}")

    # ---- Roformer hooks the orchestrator probes (separator.py:926-929) -------
    def _detect_roformer_model(self):
        """common_separator.py:521-543."""
        if not self.model_data:
            return False
        if self.model_data.get("is_roformer", False):
            return True
        for text in (self.model_path, self.model_name):
            if text and "roformer" in text.lower():
                return True
        return False

    def _initialize_roformer_loader(self):
        from .roformer_config import RoformerLoader
        self.roformer_loader = RoformerLoader()

    def get_roformer_loading_stats(self):
        return self.roformer_loader.get_loading_stats() if self.roformer_loader else {}

    def validate_roformer_config(self, config, model_type):
        return self.roformer_loader.validate_configuration(config, model_type) if self.roformer_loader else True


# Derived stem tables (class scope cannot see the table from inside a comprehension, hence here).  STEM_PAIR_MAPPER: the pairs
# that name each other's complement -- any other stem X is complemented by "No X" (CommonSeparator.secondary_stem);
# NON_ACCOM_STEMS: stems that are not an accompaniment mix (common_separator.py:49-53).
_L = CommonSeparator._STEM_LABELS
CommonSeparator.STEM_PAIR_MAPPER = {_L[a]: _L[b] for a, b in (("VOCAL_STEM", "INST_STEM"), ("LEAD_VOCAL_STEM", "BV_VOCAL_STEM"),
                                                              ("PRIMARY_STEM", "SECONDARY_STEM"))}
CommonSeparator.STEM_PAIR_MAPPER.update({v: k for k, v in list(CommonSeparator.STEM_PAIR_MAPPER.items()) if v != _L["SECONDARY_STEM"]})
CommonSeparator.NON_ACCOM_STEMS = tuple(_L[k] for k in ("VOCAL_STEM", "OTHER_STEM", "BASS_STEM", "DRUM_STEM", "GUITAR_STEM", "PIANO_STEM",
                                                         "SYNTH_STEM", "STRINGS_STEM", "WOODWINDS_STEM", "BRASS_STEM", "WIND_INST_STEM"))
del _L
