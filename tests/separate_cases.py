"""Shared body of the drop-in tests: every case of tests/golden/make_golden_separate.py (the reference's own
``separate()``) re-run through this repo's plugin classes with the same ``common_config`` / ``arch_config`` / files.

Used twice: tests/test_gpu_separate.py (libasx.so on an MI355X, the parity test proper) and
tests/test_plugin_surface.py (oracle-backed Engine double, exercising the host logic in the GPU-less container)."""
from __future__ import annotations

import hashlib
import json
import logging
import os
import random
import sys
import types
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
AUDIO = os.path.join(GOLDEN, "audio")
MODELS = os.path.join(GOLDEN, "models")
log = logging.getLogger("separate_cases")

TOL_STEM = 1e-4          # north star: stems within 1e-4 RMS (relative to the stem's RMS) of the reference CPU path


def rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


def common_config(model_name, model_path, model_data, out_dir, **over):
    """The dict Separator.load_model builds (separator.py:867-886); torch_device is what it would be on a ROCm box."""
    c = {"logger": log, "log_level": logging.WARNING, "torch_device": "cuda:0", "torch_device_cpu": "cpu", "torch_device_mps": None,
         "onnx_execution_provider": ["ROCMExecutionProvider"], "model_name": model_name, "model_path": model_path,
         "model_data": model_data, "output_format": "WAV", "output_bitrate": None, "output_dir": out_dir,
         "normalization_threshold": 0.9, "amplification_threshold": 0.0, "output_single_stem": None, "invert_using_spec": False,
         "sample_rate": 44100, "use_soundfile": False}
    c.update(over)
    return c


def load_yaml(path):
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)


# ---- model files that are regenerated from seeds (too large to commit) ---------------------------------------------
def write_demucs_repo(directory):
    """The .th packages + bag YAMLs of make_golden_separate.golden_demucs, rebuilt from the oracle's seeded weights.  ``klass``
    is pickled as ``demucs.htdemucs.HTDemucs`` (what published packages carry) through an inert stand-in class."""
    import torch
    import yaml
    from oracle import demucs_oracle as D
    os.makedirs(directory, exist_ok=True)
    cfg = D.HTConfig(channels=16, nfft=1024, depth=3, bottom_channels=128, t_layers=3, t_heads=2, samplerate=8000, segment=Fraction(1, 1))
    saved = {k: sys.modules.get(k) for k in ("demucs", "demucs.htdemucs")}
    pkg_mod, mod = types.ModuleType("demucs"), types.ModuleType("demucs.htdemucs")
    pkg_mod.__path__ = []
    klass = type("HTDemucs", (), {})
    klass.__module__ = "demucs.htdemucs"
    mod.HTDemucs = klass
    sys.modules["demucs"], sys.modules["demucs.htdemucs"] = pkg_mod, mod
    try:
        sigs = []
        for sig, seed in (("aaaa1111", 11), ("bbbb2222", 13)):
            tmp = os.path.join(directory, sig + ".th")
            torch.save({"klass": klass, "args": (), "kwargs": cfg.ctor_kwargs(), "state": D.make_ht_state(cfg, seed)}, tmp)
            with open(tmp, "rb") as f:
                h = hashlib.sha256(f.read()).hexdigest()[:8]
            os.replace(tmp, os.path.join(directory, f"{sig}-{h}.th"))
            sigs.append(sig)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    with open(os.path.join(directory, "htd_single.yaml"), "w") as f:
        yaml.safe_dump({"models": [sigs[0]]}, f)
    with open(os.path.join(directory, "htd_bag.yaml"), "w") as f:
        yaml.safe_dump({"models": sigs, "weights": [[1.0, 0.5, 2.0, 1.0], [0.5, 1.5, 1.0, 1.0]], "segment": 1}, f)
    return directory


def write_vr_model(directory, arch=31191, seed=21):
    import torch
    from oracle import vr_oracle as V
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, "vr_small_31191.pth")
    torch.save(V.make_vr_state(arch, seed), path)
    return path


# ---- case tables (mirror make_golden_separate.py) --------------------------------------------------------------------
MDX_DATA = {"compensate": 1.035, "mdx_dim_f_set": 32, "mdx_dim_t_set": 4, "mdx_n_fft_scale_set": 96, "primary_stem": "Vocals"}
MDX_ARCH = {"hop_length": 16, "segment_size": 16, "overlap": 0.25, "batch_size": 1, "enable_denoise": False}
MDXC_ARCH = {"segment_size": 12, "override_model_segment_size": False, "batch_size": 2, "overlap": 4, "pitch_shift": 0}
# "asx_res_type": the goldens were written by the reference's VRSeparator through a stand-in librosa whose resample is the
# polyphase restatement (libsamplerate is absent), so this repo's class is told to run the same converter instead of the
# platform rule (sinc_fastest on Linux); the reference ignores the key.
VR_ARCHS = {
    "vr_plain": {"batch_size": 2, "window_size": 320, "aggression": 5, "enable_tta": False, "enable_post_process": False,
                 "post_process_threshold": 0.2, "high_end_process": False, "asx_res_type": "polyphase"},
    "vr_tta_single": {"batch_size": 1, "window_size": 320, "aggression": 10, "enable_tta": True, "enable_post_process": True,
                      "post_process_threshold": 0.2, "high_end_process": True, "asx_res_type": "polyphase"},
    "vr_badsingle": {"batch_size": 4, "window_size": 512, "aggression": 5, "asx_res_type": "polyphase"},
}


def cases(family, tmp):
    """[(tag, class name, common_config, arch_config, wav, custom names)]"""
    out_dir = os.path.join(tmp, "out")
    wav = lambda n: os.path.join(AUDIO, n)  # noqa: E731
    if family == "mdx":
        mp = os.path.join(GOLDEN, "net_small.onnx")
        mk = lambda **o: common_config("net_small", mp, dict(MDX_DATA), out_dir, **o)  # noqa: E731
        return [("mdx_plain", "MDXSeparator", mk(), MDX_ARCH, wav("mdx_in.wav"), None),
                ("mdx_single", "MDXSeparator", mk(output_single_stem="instrumental"), MDX_ARCH, wav("mdx_in.wav"), None),
                ("mdx_custom", "MDXSeparator", mk(), dict(MDX_ARCH, enable_denoise=True, overlap=0.1), wav("quiet in:put?.wav"),
                 {"Vocals": "my/voc", "instrumental": "inst*"}),
                ("mdx_quiet", "MDXSeparator", mk(amplification_threshold=0.6), MDX_ARCH, wav("quiet in:put?.wav"), None)]
    if family == "mdxc":
        def mk(model, **o):
            return common_config(model, os.path.join(MODELS, model + ".ckpt"), load_yaml(os.path.join(MODELS, model + ".yaml")), out_dir, **o)
        return [("mdxc_two", "MDXCSeparator", mk("mdxc_v3two"), MDXC_ARCH, wav("mdxc_in.wav"), None),
                ("mdxc_one", "MDXCSeparator", mk("mdxc_v3one"), dict(MDXC_ARCH, overlap=2), wav("mdxc_in.wav"), None),
                ("mdxc_one_single", "MDXCSeparator", mk("mdxc_v3one", output_single_stem="Vocals"), dict(MDXC_ARCH, overlap=2),
                 wav("mdxc_in.wav"), {"vocals": "lead"})]
    if family == "roformer":
        m = "model_bs_roformer_small"
        arch = {"segment_size": 21, "override_model_segment_size": False, "batch_size": 1, "overlap": 2, "pitch_shift": 0}
        return [("rof", "MDXCSeparator", common_config(m, os.path.join(MODELS, m + ".ckpt"), load_yaml(os.path.join(MODELS, m + ".yaml")), out_dir),
                 arch, wav("rof_in.wav"), None)]
    if family == "demucs":
        repo = write_demucs_repo(os.path.join(tmp, "demucs_repo"))
        mk = lambda y, **o: common_config(y, os.path.join(repo, y + ".yaml"), {}, out_dir, sample_rate=8000, **o)  # noqa: E731
        return [("demucs_single", "DemucsSeparator", mk("htd_single"), {"segment_size": "Default", "shifts": 2, "overlap": 0.25, "segments_enabled": True},
                 wav("demucs_in.wav"), None),
                ("demucs_bag", "DemucsSeparator", mk("htd_bag", output_single_stem="Drums"),
                 {"segment_size": "Default", "shifts": 1, "overlap": 0.5, "segments_enabled": True}, wav("demucs_in.wav"), None),
                ("demucs_seg", "DemucsSeparator", mk("htd_single"), {"segment_size": "2", "shifts": 0, "overlap": 0.25, "segments_enabled": True},
                 wav("demucs_in.wav"), {"Vocals": "v", "Bass": "b"})]
    if family == "vr":
        pth = write_vr_model(os.path.join(tmp, "vr_model"))
        md = {"vr_model_param": os.path.join(MODELS, "vr_small_params"), "primary_stem": "Instrumental"}
        mk = lambda **o: common_config("vr_small_31191", pth, dict(md), out_dir, sample_rate=8000, **o)  # noqa: E731
        return [("vr_plain", "VRSeparator", mk(), VR_ARCHS["vr_plain"], wav("vr_in.wav"), None),
                ("vr_tta_single", "VRSeparator", mk(output_single_stem="Vocals"), VR_ARCHS["vr_tta_single"], wav("vr_in.wav"), None),
                ("vr_badsingle", "VRSeparator", mk(output_single_stem="Drums"), VR_ARCHS["vr_badsingle"], wav("vr_in.wav"), {"Vocals": "vv"})]
    raise KeyError(family)


def plugin_class(name):
    import importlib
    import audio_separator_amd  # noqa: F401
    mod = {"MDXSeparator": "mdx_separator", "MDXCSeparator": "mdxc_separator", "DemucsSeparator": "demucs_separator",
           "VRSeparator": "vr_separator"}[name]
    return getattr(importlib.import_module(f"audio_separator_amd.architectures.{mod}"), name)


def run_case(case, golden, monkeypatch, tol=TOL_STEM, check_files=True):
    """Instantiate like Separator.load_model, call separate() like _separate_file, compare names, arrays and files."""
    from audio_separator_amd import audio_io
    tag, cls_name, common, arch, wav, custom = case
    inst = plugin_class(cls_name)(common_config=common, arch_config=arch)
    calls = []
    real_write = inst.write_audio

    def write_audio(stem_path, stem_source):
        calls.append((stem_path, np.array(stem_source, copy=True)))
        real_write(stem_path, stem_source)
    inst.write_audio = write_audio
    if f"{tag}__offsets" in golden.files:
        draws = [int(v) for v in golden[f"{tag}__offsets"]]
        it = iter(draws)
        monkeypatch.setattr(random, "randint", lambda a, b: next(it))
    names = inst.separate(wav, custom)
    inst.clear_gpu_cache()
    want_names = json.loads(str(golden[f"{tag}__names"]))
    assert names == want_names, (tag, names, want_names)
    assert [p for p, _ in calls] == json.loads(str(golden[f"{tag}__written"]))
    worst = 0.0
    for i, (path, arr) in enumerate(calls):
        ref = golden[f"{tag}__arr{i}"]
        assert arr.shape == ref.shape, (tag, path, arr.shape, ref.shape)
        err = rel_rms(arr, ref)
        worst = max(worst, err)
        assert err < tol, (tag, path, err)
        if check_files:
            full = os.path.join(common["output_dir"], path)
            assert os.path.isfile(full), full
            pcm, sr = audio_io.read_wav(full)
            assert sr == common["sample_rate"] and pcm.shape == ref.T.shape
            # what CommonSeparator.write_audio_pydub does with the reference's array (common_separator.py:309-337)
            r = np.array(ref, np.float32, copy=True)
            peak = np.abs(r).max()
            if peak > common["normalization_threshold"]:
                r *= common["normalization_threshold"] / peak
            elif peak < common["amplification_threshold"]:
                r *= common["amplification_threshold"] / peak
            want = (r * 32767).astype(np.int16)
            got = np.rint(pcm.T * 32768.0).astype(np.int64)
            assert np.abs(got - want).max() <= max(2, int(3e-4 * 32767)), (tag, path, np.abs(got - want).max())
    inst.clear_file_specific_paths()
    assert inst.audio_file_path is None and inst.primary_source is None
    return inst, worst
