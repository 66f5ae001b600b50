"""The reference's OWN unit tests of the plugin base class, run UNMODIFIED against this repo's CommonSeparator.

`tests/ref_shims/` is put in front of the import path of a child pytest process: there `audio_separator.separator.common_separator`
resolves to the drop-in class and `soundfile` (absent from this image) to a three-function stand-in over audio_io's RIFF reader /
writer.  What runs is the reference's code: tests/unit/test_bit_depth_detection.py, tests/unit/test_bit_depth_writing.py (its
pydub / FFmpeg cases skip themselves) and the stem-swap class of tests/unit/test_stem_naming.py.  Build container only (needs
/root/reference)."""
import os
import re
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF + "/tests/unit"), reason="reference tree not present")


def test_reference_common_separator_unit_tests_pass(tmp_path):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests", "ref_shims"), ROOT])
    cmd = [sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "-c", os.devnull, "--rootdir", str(tmp_path),
           REF + "/tests/unit/test_bit_depth_detection.py", REF + "/tests/unit/test_bit_depth_writing.py",
           REF + "/tests/unit/test_stem_naming.py::TestCommonSeparatorStemSwap"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:]
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 12, tail
    assert "failed" not in tail and "error" not in tail, tail
