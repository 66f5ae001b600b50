"""Per-level comparison of the Demucs v3 engine with the oracle (GPU box): python tools/debug_hd.py [tag]"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import hdemucs_oracle as H  # noqa: E402
import audio_separator_amd as A  # noqa: E402
from test_gpu_hdemucs import demixer, ocfg  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "a"
g = np.load("tests/golden/hdemucs_small.npz")
oc = ocfg()
sd = H.make_hd_state(oc, 21)
x = g[f"x_{tag}"]
taps = {}
ref = H.hd_forward(x, sd, oc, taps)
print("oracle vs golden", float(np.abs(ref - g[f"y_{tag}"]).max()))
dm = demixer(A)
dm._load(0)
y = dm.engine.hd_forward(x)


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


def cl(t):   # NC(F)T -> channels-last
    t = t.numpy()
    return t.transpose(0, 3, 2, 1) if t.ndim == 4 else t.transpose(0, 2, 1)


D = oc.depth - 2
e = dm.engine
for i in range(D):
    r = cl(taps[f"skf{i}"])
    print(f"skf{i}", rel(e.debug_fetch(f"hd.skf{i}", r.shape), r))
    r = cl(taps[f"skt{i}"])
    print(f"skt{i}", rel(e.debug_fetch(f"hd.skt{i}", r.shape), r))
r = cl(taps["inj"])
print("inj", rel(e.debug_fetch("hd.inj", r.shape), r))
r = cl(taps[f"skf{D}"])[:, :, 0]
print("skA", rel(e.debug_fetch("hd.skA", r.shape), r))
rz = cl(taps[f"skf{D + 1}"])
print("skZ", rel(e.debug_fetch("hd.skZ", rz.shape), rz))
r2 = cl(taps["dec0"]) + r
print("dAin", rel(e.debug_fetch("hd.dAin", r2.shape), r2))
rp = cl(taps["pre1"])[:, :, 0]
print("pre", rel(e.debug_fetch("hd.pre", rp.shape), rp))
r3 = cl(taps["dec1"]) + cl(taps[f"skf{D - 1}"])
print(f"df_{D}", rel(e.debug_fetch(f"hd.df_{D}", r3.shape), r3))
r4 = cl(taps["tdec0"]) + cl(taps[f"skt{D - 1}"])
print(f"dt_{D}", rel(e.debug_fetch(f"hd.dt_{D}", r4.shape), r4))
for i in range(D - 1, -1, -1):
    j = oc.depth - 1 - i
    r = cl(taps[f"dec{j}"])
    if i > 0:
        r = r + cl(taps[f"skf{i - 1}"])
    print(f"df_{i}", rel(e.debug_fetch(f"hd.df_{i}", r.shape), r))
    r = cl(taps[f"tdec{j - 1}"])
    if i > 0:
        r = r + cl(taps[f"skt{i - 1}"])
    print(f"dt_{i}", rel(e.debug_fetch(f"hd.dt_{i}", r.shape), r))
print("out", rel(y, g[f"y_{tag}"]))
