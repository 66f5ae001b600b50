"""Host-side pieces of the fp16 x 3 Winograd weight image (csrc/kernels_wino6.h: wino6_pack_h): the float -> half conversion it
carries (round to nearest even, subnormals, carries into the exponent) against numpy's, bit for bit.  The two functions are cut out
of the header and compiled with g++ -- the header itself needs hipcc."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_half_conversion_matches_numpy(tmp_path):
    src = open(os.path.join(ROOT, "python-audio-separator_amd", "csrc", "kernels_wino6.h")).read()
    a, b = src.index("inline uint16_t wino6_f16_rne(float f)"), src.index("inline void wino6_pack_h(")
    prog = ("#include <cstdint>\n#include <cstring>\n#include <cmath>\n#include <cstdio>\n" + src[a:b] +
            "int main() { float f; while (fread(&f, 4, 1, stdin) == 1) { uint16_t h = wino6_f16_rne(f); float r = wino6_f16_f(h); "
            "fwrite(&h, 2, 1, stdout); fwrite(&r, 4, 1, stdout); } return 0; }\n")
    cpp, exe = tmp_path / "t16.cpp", tmp_path / "t16"
    cpp.write_text(prog)
    subprocess.check_call(["g++", "-O2", "-o", str(exe), str(cpp)])
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(300000) * np.exp2(rng.integers(-30, 17, 300000)),
                        [0.0, -0.0, 1.0, -1.0, 65504.0, 65519.9, 6.1e-5, 5.96e-8, 2.98e-8, 3e-8, 1e-9, 1 + 2.0 ** -11, 1 + 3 * 2.0 ** -11,
                         2.0 ** -14, 2.0 ** -24, 2.0 ** -25, 3 * 2.0 ** -25, 2047.5, 2048.5, 4095.0, 32767.9]]).astype(np.float32)
    x = x[np.abs(x) < 65520]                            # the packer scales every tile below 2^15
    out = subprocess.run([str(exe)], input=x.tobytes(), capture_output=True, check=True).stdout
    rec = np.frombuffer(out, dtype=np.dtype([("h", "<u2"), ("back", "<f4")]))
    ref = x.astype(np.float16)
    assert np.array_equal(rec["h"], ref.view(np.uint16))
    assert np.array_equal(rec["back"], ref.astype(np.float32))
