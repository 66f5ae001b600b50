"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/asx.h declares; the product path refuses to run
without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

import __graft_entry__ as entry
from audio_separator_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    entry.build()
    return E.load_library()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "asx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(asx_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"libasx.so does not export {name}"
    assert sorted(E.SYMBOLS) == declared


def test_abi_version(lib):
    hdr = open(os.path.join(ROOT, "include", "asx.h")).read()
    declared = int(re.search(r"#define\s+ASX_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert lib.asx_abi_version() == declared == E.ABI_VERSION == 7


def test_binding_refuses_a_library_of_another_abi(monkeypatch):
    """A stale libasx.so (other struct layouts) must not be driven silently: the loader compares asx_abi_version() with the
    version its ctypes structures mirror."""
    monkeypatch.setattr(E, "_lib", None)
    monkeypatch.setattr(E, "ABI_VERSION", E.ABI_VERSION + 1)
    with pytest.raises(E.AsxError, match="ABI"):
        E.load_library()
    monkeypatch.setattr(E, "_lib", None)          # the next user reloads with the real version


def test_struct_sizes_match_header():
    # 8 x 4 bytes + one double, 10 x 4 bytes (ABI 6: + norm), 6 x 8 + 4 x 4 bytes, 10 x (8 + 8 + 8 + 8)
    import ctypes as C
    assert C.sizeof(E._MdxCfg) == 40
    assert C.sizeof(E._NetCfg) == 40
    assert C.sizeof(E._Plan) == 64
    assert C.sizeof(E._Profile) == 320
    assert C.sizeof(E._LaunchRec) == 24          # struct asx_launch_rec: int32 + float + 2 doubles


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.asx_device_count() == 0
    with pytest.raises(E.AsxError):
        E.Engine(E.MDXConfig())


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "python-audio-separator_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"


def test_header_is_plain_c_and_struct_layouts_match(tmp_path):
    """include/asx.h compiles as C99 (the boundary is a C ABI, not C++), and every struct the ctypes binding mirrors has
    the size the C compiler gives it."""
    import ctypes as C
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not available")
    pairs = [("asx_mdx_config", E._MdxCfg), ("asx_net_config", E._NetCfg), ("asx_plan", E._Plan), ("asx_profile", E._Profile),
             ("asx_v3_config", E._V3Cfg), ("asx_rof_config", E._RofCfg), ("asx_ht_config", E._HtCfg), ("asx_hd_config", E._HdCfg), ("asx_vr_band", E._VrBand),
             ("asx_vr_config", E._VrCfg), ("asx_vr_params", E._VrParams), ("asx_launch_rec", E._LaunchRec)]
    src = '#include <stdio.h>\n#include "asx.h"\nint main(void) {\n' + \
          "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n, _ in pairs) + "  return 0;\n}\n"
    c = tmp_path / "sizes.c"
    c.write_text(src)
    exe = tmp_path / "sizes"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)],
                   check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, ct in pairs:
        assert int(out[name]) == C.sizeof(ct), (name, out[name], C.sizeof(ct))


def test_null_engine_is_an_error_not_a_crash(lib):
    """Every model family's begin / commit validates its engine argument before touching it (the weight-image cache flush they
    start with is per engine)."""
    for fam in ("net", "v3", "rof", "ht", "hd", "vr"):
        assert getattr(lib, f"asx_{fam}_begin")(None, None) != 0, fam
        assert getattr(lib, f"asx_{fam}_commit")(None) != 0, fam
    assert lib.asx_last_error()
