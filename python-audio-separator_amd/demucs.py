"""Host-side mirror of the reference Demucs plugin's array path (architectures/demucs_separator.py).

``DemucsDemixer(common_config, arch_config, models=[...])`` follows ``DemucsSeparator.__init__`` (:32-81) for the
arch_config keys (segment_size, shifts, overlap, segments_enabled) and ``demix_demucs`` (:162-194) for the call:
``demix(mix [2, N]) -> sources [S, 2, N]`` with stems 0/1 swapped exactly like the reference.  All sample-domain
work -- standardisation, shift trick, segment split, HTDemucs forward, triangular fold -- runs in libasx.so
(asx_ht_demix); the host only draws the random shift offsets, because the reference draws them from Python's
``random`` (uvr_lib_v5/demucs/apply.py:209).

A bag of models (BagOfModels, apply.py:26-66,169-196: htdemucs_ft, htdemucs_6s ...) is a list of
(HTConfig | HDConfig, state_dict, per-source weights); the per-model results are combined on the host with the
reference's weighted average.  HTDemucs (Demucs v4) and HDemucs (Demucs v3: `hdemucs_mmi`, asx_hd_demix) checkpoints
are accelerated; the waveform-only Demucs v1 / v2 raise NotImplementedError.
"""
from __future__ import annotations

import random

import numpy as np

from .engine import Engine, HDConfig, HTConfig, MDXConfig
from .mdx import _device_index


def htconfig_from_kwargs(kwargs: dict, max_batch: int = 0) -> HTConfig:
    """HTConfig from the `kwargs` a Demucs v4 checkpoint package stores (states.py:load_model: klass(*args, **kwargs)).
    Raises NotImplementedError for structures the engine does not build."""
    k = dict(kwargs)
    unsupported = {"cac": True, "wiener_iters": 0, "end_iters": 0, "multi_freqs": None, "dconv_mode": 1, "time_stride": 2, "context": 1, "context_enc": 0, "rewrite": True,
                   "t_emb": "sin", "t_norm_first": True, "t_norm_in": True, "t_norm_out": True, "t_layer_scale": True,
                   "t_gelu": True, "t_cross_first": False, "t_sparse_self_attn": False, "t_sparse_cross_attn": False,
                   "t_norm_in_group": False, "t_group_norm": False, "channels_time": None, "audio_channels": 2,
                   "t_weight_pos_embed": 1.0, "t_sin_random_shift": 0, "use_train_segment": True}
    for name, want in unsupported.items():
        if name in k and k[name] != want and not (want is None and not k[name]):
            raise NotImplementedError(f"HTDemucs option {name}={k[name]!r} is not built (only {want!r})")
    depth = k.get("depth", 4)
    if k.get("norm_starts", 4) < depth or k.get("dconv_attn", 4) < depth or k.get("dconv_lstm", 4) < depth:
        raise NotImplementedError("GroupNorm / attention / LSTM inside the encoder layers is not built")
    return HTConfig(sources=tuple(k["sources"]), channels=k.get("channels", 48), growth=k.get("growth", 2),
                    nfft=k.get("nfft", 4096), depth=depth, kernel_size=k.get("kernel_size", 8), stride=k.get("stride", 4),
                    dconv_depth=k.get("dconv_depth", 2), dconv_comp=k.get("dconv_comp", 8), freq_emb=k.get("freq_emb", 0.2),
                    bottom_channels=k.get("bottom_channels", 0), t_layers=k.get("t_layers", 5), t_heads=k.get("t_heads", 8),
                    t_hidden_scale=k.get("t_hidden_scale", 4.0), samplerate=k.get("samplerate", 44100),
                    segment=k.get("segment", 10), max_batch=max_batch)


def hdconfig_from_kwargs(kwargs: dict, max_batch: int = 0) -> HDConfig:
    """HDConfig from the `kwargs` of a Demucs v3 (HDemucs) checkpoint package; NotImplementedError outside the class defaults."""
    k = dict(kwargs)
    fixed = {"cac": True, "wiener_iters": 0, "end_iters": 0, "multi_freqs": None, "dconv_mode": 1, "context": 1, "context_enc": 0,
             "rewrite": True, "hybrid": True, "hybrid_old": False, "audio_channels": 2, "channels_time": None, "norm_groups": 4,
             "emb_scale": 10, "emb_smooth": True, "dconv_init": 1e-4}
    for name, want in fixed.items():
        if name in k and k[name] != want and not (want is None and not k[name]):
            raise NotImplementedError(f"HDemucs option {name}={k[name]!r} is not built (only {want!r})")
    depth = k.get("depth", 6)
    for name in ("norm_starts", "dconv_attn", "dconv_lstm"):
        if k.get(name, 4) != depth - 2:
            raise NotImplementedError(f"HDemucs {name}={k.get(name, 4)} with depth {depth}: only depth - 2 is built")
    return HDConfig(sources=tuple(k["sources"]), channels=k.get("channels", 48), growth=k.get("growth", 2), nfft=k.get("nfft", 4096),
                    depth=depth, kernel_size=k.get("kernel_size", 8), stride=k.get("stride", 4), time_stride=k.get("time_stride", 2),
                    norm_starts=depth - 2, norm_groups=4, dconv_depth=k.get("dconv_depth", 2), dconv_comp=k.get("dconv_comp", 4),
                    dconv_attn=depth - 2, dconv_lstm=depth - 2, freq_emb=k.get("freq_emb", 0.2), samplerate=k.get("samplerate", 44100),
                    segment=k.get("segment", 40), max_batch=max_batch)


def models_from_files(model_path: str, segment_size="Default", max_batch: int = 0):
    """What DemucsSeparator.separate builds before demixing (architectures/demucs_separator.py:119-124):
    ``get_demucs_model(name=<file stem>, repo=<file dir>)`` + ``demucs_segments(segment_size, model)``, as
    ([(HTConfig | HDConfig, state_dict)], weights).

    Segment rules kept from the reference: a bag's YAML ``segment`` overrides every member's (apply.py:52-53); an integer
    ``segment_size`` then overrides every member of a *bag* (apply.py:274-279) -- for a single non-bag ``.th`` the
    reference's branch raises NameError inside its bare ``try`` and leaves the model untouched (:280-292), as here."""
    import os
    from .model_files import get_demucs_model
    name = os.path.splitext(os.path.basename(model_path))[0]
    bag = get_demucs_model(name, os.path.dirname(os.path.abspath(model_path)))
    segment = bag["segment"]
    if bag["is_bag"] and segment_size not in ("Default", None):
        try:
            segment = int(segment_size)
        except (TypeError, ValueError):
            pass
    models = []
    for pkg in bag["models"]:
        kw = dict(pkg["kwargs"])
        if segment is not None:
            kw["segment"] = segment
        cfg = (htconfig_from_kwargs if pkg["kind"] == "HTDemucs" else hdconfig_from_kwargs)(kw, max_batch=max_batch)
        models.append((cfg, pkg["state"]))
    return models, bag["weights"]


def _cuda_ready() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


class DemucsDemixer:
    def __init__(self, common_config: dict, arch_config: dict, models=None, weights=None, max_batch: int = 0):
        """models: list of (HTConfig | HDConfig, state_dict); weights: per-model list of per-source weights
        (BagOfModels.weights).  Without ``models`` they are read from ``common_config["model_path"]`` (a bag ``.yaml`` or a
        ``.th`` package next to its siblings, models_from_files)."""
        if models is None:
            models, weights = models_from_files(common_config["model_path"], arch_config.get("segment_size", "Default"), max_batch)
        self.shifts = arch_config.get("shifts", 2)
        self.overlap = arch_config.get("overlap", 0.25)
        self.segments_enabled = arch_config.get("segments_enabled", True)
        self.segment_size = arch_config.get("segment_size", "Default")
        self.device = _device_index(common_config.get("torch_device", 0))
        self.models = list(models)
        if not self.models:
            raise ValueError("at least one (HTConfig, state_dict) model is required")
        S = len(self.models[0][0].sources)
        self.weights = weights if weights is not None else [[1.0] * S for _ in self.models]
        if len(self.weights) != len(self.models) or any(len(w) != S for w in self.weights):
            raise ValueError("weights must give one value per source for every model")
        self.engine = None            # the engine of the member loaded last (single models: THE engine)
        # one resident engine (weights + workspace) per bag member: htdemucs_ft keeps 4 x (108 MB of weights + the workspace of
        # `max_batch` segments, ~0.6 GB per segment at 7.8 s) = ~70 GB at the default 28 segments per pass -- sized for the
        # 288 GB of an MI355X; lower `asx_max_batch` on smaller devices.  close() releases all of it, bag buffers included.
        self.engines = [None] * len(self.models)
        self._loaded = None
        self._own = False

    def _load(self, idx: int):
        """Member ``idx`` ready on ``self.engine``.  Every member of a bag gets its OWN engine, created and committed once and
        kept for the following files (a 4-member htdemucs_ft used to re-pack and re-upload four weight sets per song); an
        engine injected from outside (``self.engine`` set by the caller: the CPU test double) is shared, members loaded in turn."""
        hc, sd = self.models[idx]
        injected = self.engine is not None and not self._own
        if injected:
            if self._loaded == idx:
                return
            eng = self.engine
        else:
            self._own = True
            if self.engines[idx] is not None:
                self.engine, self._loaded = self.engines[idx], idx
                self._demix = self.engine.hd_demix if isinstance(hc, HDConfig) else self.engine.ht_demix
                return
            # the MDX geometry of the engine is unused on this path; any valid one will do
            eng = Engine(MDXConfig(n_fft=hc.nfft, hop_length=hc.nfft // 4, dim_f=hc.nfft // 2, segment_size=8), device=self.device)
            self.engines[idx] = eng
            self.engine = eng
        if isinstance(hc, HDConfig):
            eng.load_hd(hc, sd)
            self._demix = eng.hd_demix
        else:
            eng.load_ht(hc, sd)
            self._demix = eng.ht_demix
        self._loaded = idx

    def close(self):
        for e in self.engines:
            if e is not None:
                e.close()
        self.engines = [None] * len(self.models)
        self.engine, self._loaded = None, None
        self._bag_ws = None           # three song-sized device buffers of bag_demix_dev

    def _draw_offsets(self, i, offsets):
        if not self.shifts:
            return None
        hc = self.models[i][0]
        return offsets[i] if offsets is not None else [random.randint(0, int(0.5 * hc.samplerate)) for _ in range(self.shifts)]

    def bag_demix_dev(self, mix_d, out_d, offsets=None):
        """BagOfModels with everything in HBM (``mix_d`` [2, N] -> ``out_d`` [S, 2, N], CUDA tensors): standardise once, every
        member demixes the standardised mix on its own resident engine, weighted sum / totals / de-standardise / stem swap by
        asx_ht_bag_* (apply.py:169-196, demucs_separator.py:171-189).  Only enqueues work on the current stream."""
        import torch
        st = torch.cuda.current_stream(mix_d.device).cuda_stream
        n = mix_d.shape[1]
        S = len(self.models[0][0].sources)
        ws = getattr(self, "_bag_ws", None)
        if ws is None or ws[0].shape != mix_d.shape or ws[0].device != mix_d.device:
            est = torch.empty((S, 2, n), dtype=torch.float32, device=mix_d.device)
            ws = self._bag_ws = (torch.empty_like(mix_d), est, torch.empty_like(est))
        std_d, est_d, mem_d = ws
        totals = np.zeros(S, np.float64)
        for i in range(len(self.models)):
            self._load(i)
            eng = self.engine
            if i == 0:
                eng.ht_standardize_dev(mix_d.data_ptr(), n, std_d.data_ptr(), stream=st)
            offs = self._draw_offsets(i, offsets)
            run = eng.hd_demix_dev if isinstance(self.models[i][0], HDConfig) else eng.ht_demix_dev
            run(std_d.data_ptr(), n, mem_d.data_ptr(), shifts=self.shifts, offsets=offs, overlap=self.overlap, flags=0, stream=st)
            eng.ht_bag_accumulate_dev(est_d.data_ptr(), mem_d.data_ptr(), self.weights[i], n, first=(i == 0), stream=st)
            totals += np.asarray(self.weights[i], np.float64)
        self.engine.ht_bag_finish_dev(est_d.data_ptr(), totals.astype(np.float32), mix_d.data_ptr(), n, out_d.data_ptr(),
                                      standardize=True, swap01=True, stream=st)

    def demix_dev(self, mix_d, offsets=None):
        """demix_demucs with input and output in HBM: CUDA tensor [2, N] -> CUDA tensor [S, 2, N] (sources 0 / 1 swapped), or None
        when this configuration has no device-resident path (``segments_enabled=False`` windows on the host)."""
        import torch
        if not self.segments_enabled:
            return None
        n = mix_d.shape[1]
        out_d = torch.empty((len(self.models[0][0].sources), 2, n), dtype=torch.float32, device=mix_d.device)
        if len(self.models) > 1:
            self.bag_demix_dev(mix_d, out_d, offsets)
            return out_d
        self._load(0)
        offs = self._draw_offsets(0, offsets)
        run = self.engine.hd_demix_dev if isinstance(self.models[0][0], HDConfig) else self.engine.ht_demix_dev
        run(mix_d.data_ptr(), n, out_d.data_ptr(), shifts=self.shifts, offsets=offs, overlap=self.overlap, flags=3,
            stream=torch.cuda.current_stream(mix_d.device).cuda_stream)
        return out_d

    def _bag_on_device(self, mix: np.ndarray, offsets):
        """One upload of the mix, one download of the result around bag_demix_dev."""
        import torch
        dev = torch.device("cuda", self.device)
        mix_d = torch.from_numpy(mix).to(dev)
        out_d = torch.empty((len(self.models[0][0].sources), 2, mix.shape[1]), dtype=torch.float32, device=dev)
        self.bag_demix_dev(mix_d, out_d, offsets)
        return out_d.cpu().numpy()

    def demix(self, mix: np.ndarray, offsets=None) -> np.ndarray:
        """demix_demucs: [2, N] -> [S, 2, N].  offsets (optional) pins the shift draws, one list per model."""
        mix = np.ascontiguousarray(mix, np.float32)
        if mix.ndim != 2 or mix.shape[0] != 2:
            raise ValueError(f"Expected a 2-channel audio signal, but got shape {mix.shape}")
        est = None
        totals = np.zeros(len(self.models[0][0].sources), np.float64)
        single = len(self.models) == 1 and self.segments_enabled   # split=False combines on the host like a bag
        if not single and self.segments_enabled and (self._own or self.engine is None) and _cuda_ready():
            return self._bag_on_device(mix, offsets)
        for i in range(len(self.models)):
            self._load(i)
            offs = self._draw_offsets(i, offsets)
            # a single model is de-standardised and swapped inside the engine; a bag is combined first (apply.py:186-196)
            out = self._demix(mix, shifts=self.shifts, offsets=offs, overlap=self.overlap, standardize=single,
                              swap01=single) if single else self._bag_member(mix, offs)
            if single:
                return out
            w = np.asarray(self.weights[i], np.float32)
            out *= w[:, None, None]
            totals += w
            est = out if est is None else est + out
        est /= totals.astype(np.float32)[:, None, None]
        ref = mix.mean(0)
        import torch
        rt = torch.from_numpy(ref)
        est = est * float(rt.std()) + float(rt.mean())
        est[[0, 1]] = est[[1, 0]]
        return est

    def _apply_whole(self, std_mix: np.ndarray, hc, offs) -> np.ndarray:
        """apply_model(..., split=False) (apply.py:198-214, 251-260): one forward per shift over the whole (shifted) track.
        HDemucs runs it at its own length; HTDemucs centres it in a training-length window (valid_length) and trims."""
        n = std_mix.shape[1]
        max_shift = int(0.5 * hc.samplerate) if self.shifts else 0
        padded = np.pad(std_mix, ((0, 0), (max_shift, max_shift)))
        out = None
        for off in (offs if self.shifts else [0]):
            length = n + max_shift - off
            view = padded[:, off:off + length]
            if isinstance(hc, HDConfig):
                y = self.engine.hd_forward(view[None])[0]
            else:
                tl = hc.segment_samples
                if length > tl:
                    raise ValueError(f"segments_enabled=False: {length} samples exceed the model's training length {tl} "
                                     "(the reference fails inside HTDemucs for such inputs)")
                delta = tl - length
                start = off - delta // 2                      # TensorChunk.padded: real context from the padded mix
                lo, hi = max(0, start), min(padded.shape[1], start + tl)
                win = np.zeros((2, tl), np.float32)
                win[:, lo - start:hi - start] = padded[:, lo:hi]
                y = self.engine.ht_forward(win[None])[0][..., delta // 2: delta // 2 + length]
            y = y[..., max_shift - off:]
            out = y if out is None else out + y
        return out / max(self.shifts, 1)

    def _bag_member(self, mix, offs):
        import torch
        t = torch.from_numpy(mix)
        ref = t.mean(0)
        std_mix = ((t - ref.mean()) / ref.std()).numpy()
        if not self.segments_enabled:
            return self._apply_whole(std_mix, self.models[self._loaded][0], offs)
        return self._demix(std_mix, shifts=self.shifts, offsets=offs, overlap=self.overlap)
