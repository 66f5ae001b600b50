"""CPU run of the plugin-surface classes (architectures/*.py) against the reference's own ``separate()`` goldens, with the
Engine replaced by the oracle-backed double of tests/fake_engine.py: checks everything ABOVE the C ABI -- constructor
contract, model-file readers, configuration mirrors, per-file rules (short-audio segment override, output_single_stem,
custom names, sanitising), what reaches the writer, the WAV files -- in the GPU-less container.  The same cases run
against libasx.so in tests/test_gpu_separate.py."""
import os

import numpy as np
import pytest

from tests import fake_engine, separate_cases as SC


@pytest.fixture()
def golden(request):
    return np.load(os.path.join(SC.GOLDEN, f"separate_{request.param}.npz"))


def _run(family, tmp_path, monkeypatch, tol=2e-5):
    fake_engine.install(monkeypatch)
    g = np.load(os.path.join(SC.GOLDEN, f"separate_{family}.npz"))
    res = []
    for case in SC.cases(family, str(tmp_path)):
        inst, worst = SC.run_case(case, g, monkeypatch, tol=tol)
        res.append((case[0], inst, worst))
    assert fake_engine.OracleEngine.created >= 1
    return g, res


def test_mdx(tmp_path, monkeypatch):
    _, res = _run("mdx", tmp_path, monkeypatch)
    inst = res[0][1]
    assert (inst.n_bins, inst.trim, inst.chunk_size, inst.gen_size) == (49, 48, 240, 144)     # initialize_model_settings
    assert inst.primary_stem_name == "Vocals" and inst.secondary_stem_name == "Instrumental"


def test_mdxc_tfc(tmp_path, monkeypatch):
    g, res = _run("mdxc", tmp_path, monkeypatch)
    for tag, inst, _ in res:
        assert bool(inst.override_model_segment_size) == bool(g[f"{tag}__override"])          # < 10 s rule (mdxc_separator.py:131-138)


def test_roformer(tmp_path, monkeypatch):
    g, res = _run("roformer", tmp_path, monkeypatch)
    inst = res[0][1]
    assert inst.is_roformer_model is True
    import json
    assert inst.get_roformer_loading_stats() == json.loads(str(g["rof__stats"]))


def test_demucs(tmp_path, monkeypatch):
    _run("demucs", tmp_path, monkeypatch, tol=5e-5)


def test_vr(tmp_path, monkeypatch):
    _run("vr", tmp_path, monkeypatch, tol=5e-5)
