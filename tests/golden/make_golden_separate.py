#!/usr/bin/env python3
"""Golden vectors for the FILE-LEVEL plugin surface: the reference's own ``separate(audio_file_path, custom_output_names)``
of all four architecture classes, constructed exactly like ``Separator.load_model`` constructs them
(``separator_class(common_config=..., arch_config=...)``, separator.py:889-914), run on real files here in the build
container (needs /root/reference).

What is real: MDXSeparator / MDXCSeparator / DemucsSeparator / VRSeparator and everything under them (CommonSeparator,
spec_utils, STFT, ConvTDFNet, TFC_TDF_net, RoformerLoader + BSRoformer, pretrained.get_model + states.load_model +
BagOfModels + apply_model + HTDemucs, nets.determine_model_capacity), the model FILES (.onnx / .ckpt / .th + .yaml /
.pth, written below with the reference's classes and then loaded by the reference through its normal path) and the
input WAV files.
What is stubbed, because the package is absent from this image: ``librosa.load`` -> stdlib ``wave`` reader of the PCM16
inputs (sample / 32768, what libsndfile yields); ``soundfile.info``; ``onnxruntime.InferenceSession`` -> the torch
ConvTDFNet the .onnx file was exported from (tests/golden/make_onnx_fixture.py); ``ml_collections.ConfigDict``,
``beartype``, ``rotary_embedding_torch`` (restated in oracle/roformer_oracle.py), ``julius``, ``diffq``; ``librosa.stft /
istft / resample`` for VR -> oracle/vr_oracle.py restatements with the reference's ARM resampler setting (polyphase),
as in make_golden_vr.py.  ``write_audio`` is replaced on the instance by a recorder of ``(stem_path, ndarray)``:
encode is out of scope; what is compared is the array each class hands to the writer, and the returned names.

    python tests/golden/make_golden_separate.py        # writes separate_*.npz, audio/*.wav, models/*
"""
import importlib.machinery
import json
import logging
import os
import random
import sys
import types
import typing
import wave as wave_mod
from fractions import Fraction

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
AUDIO = os.path.join(HERE, "audio")
MODELS = os.path.join(HERE, "models")
BIG = "/tmp/asx_golden_models"          # regenerated from seeds by the tests; too large to commit
sys.path.insert(0, ROOT)
from oracle import demucs_oracle as D  # noqa: E402
from oracle import mdx_oracle as O  # noqa: E402
from oracle import mdxc_oracle as M  # noqa: E402
from oracle import roformer_oracle as R  # noqa: E402
from oracle import vr_oracle as V  # noqa: E402

log = logging.getLogger("golden")


class ConfigDict(dict):
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = ConfigDict(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def read_pcm16(path):
    with wave_mod.open(path, "rb") as w:
        assert w.getsampwidth() == 2
        ch, sr, n = w.getnchannels(), w.getframerate(), w.getnframes()
        x = np.frombuffer(w.readframes(n), "<i2").reshape(-1, ch).T.astype(np.float32) / 32768.0
    return np.ascontiguousarray(x), sr


def write_pcm16(path, x, sr):
    """x float [2, n] in (-1, 1) -> PCM16 WAV; returns the decoded float32 signal the readers will see."""
    q = np.clip(np.rint(x * 32768.0), -32768, 32767).astype("<i2")
    with wave_mod.open(path, "wb") as w:
        w.setnchannels(q.shape[0])
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(np.ascontiguousarray(q.T).tobytes())
    return q.astype(np.float32) / 32768.0


def librosa_load(path, sr=22050, mono=True, **kw):
    x, file_sr = read_pcm16(path)
    assert sr is None or sr == file_sr, (sr, file_sr)
    if x.shape[0] == 1:
        x = x[0]
    return x, file_sr


class SfInfo:
    subtype = "PCM_16"


class FakeOrtSession:
    """onnxruntime.InferenceSession for the committed .onnx fixtures: runs the torch ConvTDFNet they were exported from."""
    nets = {}

    def __init__(self, path, providers=None, sess_options=None):
        self.net = FakeOrtSession.nets[os.path.basename(path)]

    def run(self, _, feed):
        with torch.no_grad():
            return [self.net(torch.as_tensor(feed["input"], dtype=torch.float32)).numpy()]


def install_stubs():
    for name in ["onnx", "onnx2torch", "audioread", "julius"]:
        _stub(name)
    _stub("onnxruntime", InferenceSession=FakeOrtSession, SessionOptions=lambda: types.SimpleNamespace(log_severity_level=0))
    _stub("librosa", load=librosa_load, stft=V.lr_stft, istft=V.lr_istft, resample=V.lr_resample,
          get_duration=lambda **k: 0.0)
    _stub("soundfile", info=lambda p: SfInfo())
    _stub("pydub", AudioSegment=object)
    _stub("pytorch_lightning", LightningModule=torch.nn.Module)
    _stub("ml_collections", ConfigDict=ConfigDict)
    _stub("beartype", beartype=lambda f: f)
    _stub("beartype.typing", Tuple=typing.Tuple, Optional=typing.Optional, List=typing.List, Callable=typing.Callable)
    _stub("rotary_embedding_torch", RotaryEmbedding=R.RotaryEmbedding)
    _stub("diffq", UniformQuantizer=object, DiffQuantizer=object, restore_quantized_state=lambda *a, **k: None)
    try:
        import six  # noqa: F401
    except ImportError:
        _stub("six", PY2=False)
    for name, path in [("audio_separator", REF + "/audio_separator"), ("audio_separator.separator", REF + "/audio_separator/separator")]:
        pkg = types.ModuleType(name)
        pkg.__path__ = [path]
        pkg.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = pkg


def common_config(model_name, model_path, model_data, out_dir, **over):
    """The dict Separator.load_model builds (separator.py:867-886)."""
    c = {"logger": log, "log_level": logging.WARNING, "torch_device": torch.device("cpu"), "torch_device_cpu": torch.device("cpu"),
         "torch_device_mps": None, "onnx_execution_provider": ["CPUExecutionProvider"], "model_name": model_name,
         "model_path": model_path, "model_data": model_data, "output_format": "WAV", "output_bitrate": None, "output_dir": out_dir,
         "normalization_threshold": 0.9, "amplification_threshold": 0.0, "output_single_stem": None, "invert_using_spec": False,
         "sample_rate": 44100, "use_soundfile": False}
    c.update(over)
    return c


def record(instance):
    """Replace write_audio by a recorder; returns the list it appends (stem_path, array copy) to."""
    calls = []
    instance.write_audio = lambda stem_path, stem_source: calls.append((stem_path, np.array(stem_source, copy=True)))
    return calls


def pack(out, tag, names, calls):
    out[f"{tag}__names"] = np.array(json.dumps(list(names)))
    out[f"{tag}__written"] = np.array(json.dumps([p for p, _ in calls]))
    for i, (_, a) in enumerate(calls):
        out[f"{tag}__arr{i}"] = np.asarray(a, np.float32)


def synth(seed, n, sr, amp=0.4):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    x = np.stack([np.sin(2 * np.pi * 0.013 * sr * t) * 0.5 + rng.standard_normal(n) * 0.3,
                  np.sin(2 * np.pi * 0.021 * sr * t + 1.0) * 0.4 + rng.standard_normal(n) * 0.3])
    return (amp * x / np.abs(x).max()).astype(np.float64)


# ---------------------------------------------------------------------------------------------------------------
def golden_mdx(out):
    from audio_separator.separator.architectures.mdx_separator import MDXSeparator
    from audio_separator.separator.uvr_lib_v5.mdxnet import ConvTDFNet
    for fname, bias, seed in [("net_small.onnx", False, 3)]:
        d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4, bias=bias)
        net = ConvTDFNet("t", 1e-3, "rmsprop", 4, 32, 16, 96, 16, 5, 2, 8, 3, 4, bias, 0)
        net.load_state_dict(O.make_convtdf_state(d, seed=seed), strict=False)
        FakeOrtSession.nets[fname] = net.eval()
    wav = os.path.join(AUDIO, "mdx_in.wav")
    write_pcm16(wav, synth(201, 3000, 44100, amp=0.97), 44100)       # peak > 0.9: normalize() scales
    quiet = os.path.join(AUDIO, "quiet in:put?.wav")                   # exercises sanitize_filename
    write_pcm16(quiet, synth(202, 1500, 44100, amp=0.3), 44100)
    model_data = {"compensate": 1.035, "mdx_dim_f_set": 32, "mdx_dim_t_set": 4, "mdx_n_fft_scale_set": 96, "primary_stem": "Vocals"}
    arch = {"hop_length": 16, "segment_size": 16, "overlap": 0.25, "batch_size": 1, "enable_denoise": False}
    mp = os.path.join(HERE, "net_small.onnx")
    cases = [("mdx_plain", wav, {}, arch, None),
             ("mdx_single", wav, {"output_single_stem": "instrumental"}, arch, None),
             ("mdx_custom", quiet, {}, dict(arch, enable_denoise=True, overlap=0.1), {"Vocals": "my/voc", "instrumental": "inst*"}),
             ("mdx_quiet", quiet, {"amplification_threshold": 0.6}, arch, None)]
    for tag, path, over, a, names in cases:
        s = MDXSeparator(common_config=common_config("net_small", mp, model_data, "/tmp/out", **over), arch_config=a)
        calls = record(s)
        pack(out, tag, s.separate(path, names), calls)


def golden_mdxc(out):
    from audio_separator.separator.architectures.mdxc_separator import MDXCSeparator
    from audio_separator.separator.uvr_lib_v5.tfc_tdf_v3 import TFC_TDF_net
    cfg2 = M.V3Config(n_fft=128, hop_length=16, dim_f=64, dim_t=16, num_subbands=2, num_scales=2, num_blocks_per_scale=2,
                      num_channels_model=8, growth=8, bottleneck_factor=4)
    cfg1 = M.V3Config(n_fft=128, hop_length=16, dim_f=64, dim_t=16, num_subbands=2, num_scales=2, num_blocks_per_scale=1,
                      num_channels_model=8, growth=4, bottleneck_factor=2, target_instrument="Vocals", act="relu")
    wav = os.path.join(AUDIO, "mdxc_in.wav")
    write_pcm16(wav, synth(211, 3000, 44100, amp=0.95), 44100)
    for tag, cfg, seed in (("v3two", cfg2, 5), ("v3one", cfg1, 6)):
        net = TFC_TDF_net(ConfigDict(cfg.as_model_data()), device=torch.device("cpu"))
        net.load_state_dict(M.make_v3_state(cfg, seed), strict=True)
        path = os.path.join(MODELS, f"mdxc_{tag}.ckpt")
        torch.save(net.state_dict(), path)
        with open(os.path.join(MODELS, f"mdxc_{tag}.yaml"), "w") as f:
            yaml.safe_dump(cfg.as_model_data(), f)
    md2, md1 = cfg2.as_model_data(), cfg1.as_model_data()
    # 3000 samples at 44.1 kHz is < 10 s: the class flips override_model_segment_size itself and uses arch segment_size
    arch = {"segment_size": 12, "override_model_segment_size": False, "batch_size": 2, "overlap": 4, "pitch_shift": 0}
    cases = [("mdxc_two", "mdxc_v3two", md2, {}, arch, None),
             ("mdxc_one", "mdxc_v3one", md1, {}, dict(arch, overlap=2), None),
             ("mdxc_one_single", "mdxc_v3one", md1, {"output_single_stem": "Vocals"}, dict(arch, overlap=2), {"vocals": "lead"})]
    for tag, model, md, over, a, names in cases:
        s = MDXCSeparator(common_config=common_config(model, os.path.join(MODELS, model + ".ckpt"), md, "/tmp/out", **over), arch_config=a)
        calls = record(s)
        pack(out, tag, s.separate(wav, names), calls)
        out[f"{tag}__override"] = np.array(bool(s.override_model_segment_size))


def golden_roformer(out):
    from audio_separator.separator.architectures.mdxc_separator import MDXCSeparator
    from audio_separator.separator.uvr_lib_v5.roformer.bs_roformer import BSRoformer
    cfg = R.RoformerConfig(dim=32, depth=2, heads=2, dim_head=64, freqs_per_bands=(2, 2, 4, 8, 17), stft_n_fft=64,
                           stft_hop_length=16, stft_win_length=64, dim_t=21, sample_rate=100, mlp_expansion_factor=2)
    net = BSRoformer(**cfg.model_kwargs(), flash_attn=False)
    net.load_state_dict(R.make_roformer_state(cfg, 7), strict=True)
    path = os.path.join(MODELS, "model_bs_roformer_small.ckpt")
    torch.save(net.state_dict(), path)
    md = cfg.as_model_data()
    with open(os.path.join(MODELS, "model_bs_roformer_small.yaml"), "w") as f:
        yaml.safe_dump(json.loads(json.dumps(md)), f)
    wav = os.path.join(AUDIO, "rof_in.wav")
    write_pcm16(wav, synth(221, 1000, 44100, amp=0.8), 44100)
    arch = {"segment_size": 21, "override_model_segment_size": False, "batch_size": 1, "overlap": 2, "pitch_shift": 0}
    s = MDXCSeparator(common_config=common_config("model_bs_roformer_small", path, md, "/tmp/out"), arch_config=arch)
    calls = record(s)
    pack(out, "rof", s.separate(wav, None), calls)
    out["rof__stats"] = np.array(json.dumps(s.get_roformer_loading_stats()))
    out["rof__is_roformer"] = np.array(bool(s.is_roformer_model))


def golden_demucs(out):
    from audio_separator.separator.architectures.demucs_separator import DemucsSeparator
    # published packages pickle `demucs.htdemucs.HTDemucs`; DemucsSeparator.__init__ puts uvr_lib_v5 on sys.path for that
    sys.path.insert(0, os.path.join(REF, "audio_separator", "separator", "uvr_lib_v5"))
    from demucs.htdemucs import HTDemucs
    cfg = D.HTConfig(channels=16, nfft=1024, depth=3, bottom_channels=128, t_layers=3, t_heads=2, samplerate=8000, segment=Fraction(1, 1))
    import hashlib
    sigs = []
    for sig, seed in (("aaaa1111", 11), ("bbbb2222", 13)):
        model = HTDemucs(**cfg.ctor_kwargs())
        model.load_state_dict(D.make_ht_state(cfg, seed))
        pkg = {"klass": HTDemucs, "args": (), "kwargs": cfg.ctor_kwargs(), "state": {k: v.clone() for k, v in model.state_dict().items()}}
        tmp = os.path.join(BIG, "demucs", sig + ".th")
        torch.save(pkg, tmp)
        h = hashlib.sha256(open(tmp, "rb").read()).hexdigest()[:8]
        os.replace(tmp, os.path.join(BIG, "demucs", f"{sig}-{h}.th"))
        sigs.append(sig)
    with open(os.path.join(BIG, "demucs", "htd_single.yaml"), "w") as f:
        yaml.safe_dump({"models": [sigs[0]]}, f)
    with open(os.path.join(BIG, "demucs", "htd_bag.yaml"), "w") as f:
        yaml.safe_dump({"models": sigs, "weights": [[1.0, 0.5, 2.0, 1.0], [0.5, 1.5, 1.0, 1.0]], "segment": 1}, f)
    wav = os.path.join(AUDIO, "demucs_in.wav")
    L = int(2.6 * 8000) + 123
    write_pcm16(wav, synth(231, L, 8000, amp=0.7), 8000)
    cases = [("demucs_single", "htd_single.yaml", {"segment_size": "Default", "shifts": 2, "overlap": 0.25, "segments_enabled": True}, {}, None),
             ("demucs_bag", "htd_bag.yaml", {"segment_size": "Default", "shifts": 1, "overlap": 0.5, "segments_enabled": True},
              {"output_single_stem": "Drums"}, None),
             ("demucs_seg", "htd_single.yaml", {"segment_size": "2", "shifts": 0, "overlap": 0.25, "segments_enabled": True}, {},
              {"Vocals": "v", "Bass": "b"})]
    for tag, yml, arch, over, names in cases:
        s = DemucsSeparator(common_config=common_config(os.path.splitext(yml)[0], os.path.join(BIG, "demucs", yml), {}, "/tmp/out",
                                                        sample_rate=8000, **over), arch_config=arch)
        calls = record(s)
        random.seed(4321)
        offs = []
        real = random.randint

        def rec(a, b):
            offs.append(real(a, b))
            return offs[-1]
        random.randint = rec
        try:
            names_out = s.separate(wav, names)
        finally:
            random.randint = real
        pack(out, tag, names_out, calls)
        out[f"{tag}__offsets"] = np.array(offs, np.int64)


def golden_vr(out):
    from audio_separator.separator.architectures.vr_separator import VRSeparator
    from audio_separator.separator.uvr_lib_v5 import spec_utils
    spec_utils.wav_resolution = "polyphase"      # the reference's macOS-ARM / MPS synthesis setting (spec_utils.py:33)
    params = dict(V.small_params().param)
    params["sr"] = 44100      # == 44100: no final librosa.resample (soxr_hq, a host library) after spec_to_wav (vr_separator.py:218)
    pj = os.path.join(MODELS, "vr_small_params.json")
    with open(pj, "w") as f:
        json.dump({k: ({str(d): b for d, b in v.items()} if k == "band" else v) for k, v in params.items()
                   if k in ("bins", "unstable_bins", "reduction_bins", "band", "sr", "pre_filter_start", "pre_filter_stop")}, f, indent=1)
    arch_id, seed = 31191, 21
    sd = V.make_vr_state(arch_id, seed)
    pth = os.path.join(BIG, "vr_small_31191.pth")
    torch.save(sd, pth)
    size_kb = int(np.ceil(os.stat(pth).st_size / 1024))
    print("VR .pth size KB", size_kb)
    n = 8000 * 3 + 137
    wav = os.path.join(AUDIO, "vr_in.wav")
    write_pcm16(wav, synth(241, n, 8000, amp=0.6), 8000)
    md = {"vr_model_param": os.path.splitext(pj)[0], "primary_stem": "Instrumental"}
    for tag, over, arch, names in (
            ("vr_plain", {}, {"batch_size": 2, "window_size": 320, "aggression": 5, "enable_tta": False, "enable_post_process": False,
                              "post_process_threshold": 0.2, "high_end_process": False}, None),
            ("vr_tta_single", {"output_single_stem": "Vocals"}, {"batch_size": 1, "window_size": 320, "aggression": 10, "enable_tta": True,
                                                                  "enable_post_process": True, "post_process_threshold": 0.2,
                                                                  "high_end_process": True}, None),
            ("vr_badsingle", {"output_single_stem": "Drums"}, {"batch_size": 4, "window_size": 512, "aggression": 5}, {"Vocals": "vv"})):
        s = VRSeparator(common_config=common_config("vr_small_31191", pth, md, "/tmp/out", sample_rate=8000, **over), arch_config=arch)
        calls = record(s)
        pack(out, tag, s.separate(wav, names), calls)
    out["vr__arch"] = np.array(arch_id)
    out["vr__seed"] = np.array(seed)


def main():
    logging.basicConfig(level=logging.ERROR)
    torch.set_num_threads(os.cpu_count())
    for d in (AUDIO, MODELS, BIG, os.path.join(BIG, "demucs")):
        os.makedirs(d, exist_ok=True)
    install_stubs()
    only = sys.argv[1:]
    for name, fn in (("mdx", golden_mdx), ("mdxc", golden_mdxc), ("roformer", golden_roformer), ("demucs", golden_demucs),
                     ("vr", golden_vr)):
        if only and name not in only:
            continue
        out = {}
        fn(out)
        path = os.path.join(HERE, f"separate_{name}.npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path), {k: (v.shape if v.ndim else str(v)[:80]) for k, v in out.items()})


if __name__ == "__main__":
    main()
