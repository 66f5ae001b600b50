"""The reference's OWN orchestrator (audio_separator/separator/separator.py: Separator.load_model -> separate ->
_separate_file) driving this repo's plugin classes after ``audio_separator_amd.install()`` -- not one line of the
reference modified.  Build container only (needs /root/reference; skipped elsewhere); the Engine is the oracle-backed
double, because there is no GPU here: what this proves is the plugin CONTRACT (how L4 resolves, constructs, calls and
cleans up a model instance, separator.py:889-914, 1027-1036), the numbers are proven on the GPU in test_gpu_separate.py.
Third-party packages absent from this image are stubbed; model download and the hash -> model_data lookup (network) are
replaced by local answers, like the reference's own unit tests do."""
import importlib
import importlib.machinery
import json
import os
import sys
import types

import numpy as np
import pytest

from tests import fake_engine, separate_cases as SC

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF + "/audio_separator"), reason="reference tree not present")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


@pytest.fixture()
def reference_separator(monkeypatch):
    mods = {n: _stub(n) for n in ("onnx", "onnx2torch", "audioread", "soundfile", "librosa")}
    mods["onnxruntime"] = _stub("onnxruntime", get_available_providers=lambda: [])
    mods["pydub"] = _stub("pydub", AudioSegment=object)
    for name, path in (("audio_separator", REF + "/audio_separator"), ("audio_separator.separator", REF + "/audio_separator/separator")):
        pkg = types.ModuleType(name)
        pkg.__path__ = [path]
        pkg.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        mods[name] = pkg
    for k, v in mods.items():
        monkeypatch.setitem(sys.modules, k, v)
    for k in [k for k in sys.modules if k.startswith("audio_separator.separator.")]:
        monkeypatch.delitem(sys.modules, k)
    sep_mod = importlib.import_module("audio_separator.separator.separator")
    yield sep_mod.Separator
    for k in [k for k in sys.modules if k.startswith("audio_separator.")]:
        sys.modules.pop(k, None)


def test_reference_separator_runs_our_mdx_plugin(reference_separator, tmp_path, monkeypatch):
    import audio_separator_amd as A
    fake_engine.install(monkeypatch)
    names = A.install()
    try:
        assert "audio_separator.separator.architectures.mdx_separator" in names
        out_dir = str(tmp_path / "out")
        sep = reference_separator(info_only=True, output_dir=out_dir, model_file_dir=str(tmp_path / "models"),
                                  mdx_params=dict(SC.MDX_ARCH))
        sep.torch_device, sep.torch_device_cpu, sep.onnx_execution_provider = "cuda:0", "cpu", ["ROCMExecutionProvider"]
        model_path = os.path.join(SC.GOLDEN, "net_small.onnx")
        monkeypatch.setattr(sep, "download_model_files", lambda f: (f, "MDX", "small test net", model_path, None))
        monkeypatch.setattr(sep, "load_model_data_using_hash", lambda p: dict(SC.MDX_DATA))
        sep.load_model("net_small.onnx")
        inst = sep.model_instance
        assert type(inst).__module__ == "audio_separator_amd.architectures.mdx_separator"      # ours, resolved by the reference
        files = sep.separate(os.path.join(SC.AUDIO, "mdx_in.wav"))
        g = np.load(os.path.join(SC.GOLDEN, "separate_mdx.npz"))
        assert files == json.loads(str(g["mdx_plain__names"]))
        for f in files:
            assert os.path.isfile(os.path.join(out_dir, f))
        assert inst.audio_file_path is None and inst.primary_source is None                    # clear_file_specific_paths ran
        # custom names + a second file through the same instance (the object is reused across files, separator.py:1027-1036)
        files2 = sep.separate(os.path.join(SC.AUDIO, "quiet in:put?.wav"), {"Vocals": "v2"})
        assert files2 == ["quiet in_put_(Instrumental)_net_small.wav", "v2.wav"]
    finally:
        A.uninstall()


def test_reference_separator_runs_our_mdxc_roformer_plugin(reference_separator, tmp_path, monkeypatch):
    import audio_separator_amd as A
    fake_engine.install(monkeypatch)
    A.install()
    try:
        m = "model_bs_roformer_small"
        sep = reference_separator(info_only=True, output_dir=str(tmp_path / "out"), model_file_dir=str(tmp_path / "models"),
                                  mdxc_params={"segment_size": 21, "override_model_segment_size": False, "batch_size": 1, "overlap": 2,
                                               "pitch_shift": 0})
        sep.torch_device, sep.torch_device_cpu = "cuda:0", "cpu"
        ckpt, yml = os.path.join(SC.MODELS, m + ".ckpt"), os.path.join(SC.MODELS, m + ".yaml")
        monkeypatch.setattr(sep, "download_model_files", lambda f: (f, "MDXC", "small roformer", ckpt, yml))
        sep.load_model(m + ".ckpt")                       # reads the YAML with the reference's own load_model_data_from_yaml
        assert sep.model_instance.is_roformer_model and sep.model_instance.get_roformer_loading_stats()["new_implementation_success"] == 1
        files = sep.separate(os.path.join(SC.AUDIO, "rof_in.wav"))
        g = np.load(os.path.join(SC.GOLDEN, "separate_roformer.npz"))
        assert files == json.loads(str(g["rof__names"]))
    finally:
        A.uninstall()


def test_stem_vocabulary_equals_the_references():
    """Every upper-case class constant of the reference's CommonSeparator (stem names, STEM_PAIR_MAPPER, NON_ACCOM_STEMS,
    common_separator.py:19-53) has the same value here: the orchestrator, the presets and the file names rely on them."""
    import ast

    import audio_separator_amd as A
    tree = ast.parse(open(REF + "/audio_separator/separator/common_separator.py").read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "CommonSeparator")
    ns = {}
    for node in cls.body:
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and node.targets[0].id.isupper():
            exec(compile(ast.Module([node], []), "ref_constants", "exec"), ns)
    consts = {k: v for k, v in ns.items() if k.isupper()}
    assert len(consts) >= 30
    for k, v in consts.items():
        assert getattr(A.CommonSeparator, k) == v, k
