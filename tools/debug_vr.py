import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import vr_oracle as V
import audio_separator_amd as A
from test_gpu_vr import demixer, SMALL_CAP

g = np.load("tests/golden/vr_small.npz")
arch, seed = 123821, 5
dm = demixer(A, arch, seed)
sd = V.make_vr_state(arch, seed, SMALL_CAP)
x = g["hp_net_in"]
y = dm.engine.vr_forward(x)
B = x.shape[0]
F_, W = 96, 64
ct = 12
hc = dm.engine.debug_fetch("vr.hc", (B, F_, W, ct))
xt = torch.tensor(x)[:, :, :96]
bw = 48
def rel(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))
print("x slot", rel(hc[..., :2], xt.permute(0, 2, 3, 1).numpy()), "pad", np.abs(hc[..., 2:4]).max())
aux1 = torch.cat([V._base(xt[:, :, :bw], sd, "stg1_low_band_net", arch), V._base(xt[:, :, bw:], sd, "stg1_high_band_net", arch)], dim=2)
print("aux1 low", rel(hc[:, :bw, :, 4:8], aux1[:, :, :bw].permute(0, 2, 3, 1).numpy()), "high", rel(hc[:, bw:, :, 4:8], aux1[:, :, bw:].permute(0, 2, 3, 1).numpy()))
h = torch.cat([xt, aux1], dim=1)
y2 = V._cba(h, sd, "stg2_bridge", 1, 0)
print("y2", rel(dm.engine.debug_fetch("vr.y2", (B, F_, W, 4)), y2.permute(0, 2, 3, 1).numpy()))
aux2 = V._base(y2, sd, "stg2_full_band_net", arch)
print("aux2", rel(hc[..., 8:12], aux2.permute(0, 2, 3, 1).numpy()))
h = torch.cat([xt, aux1, aux2], dim=1)
y3 = V._cba(h, sd, "stg3_bridge", 1, 0)
print("y3", rel(dm.engine.debug_fetch("vr.y3", (B, F_, W, 4)), y3.permute(0, 2, 3, 1).numpy()))
# stage-3 internals
p = "stg3_full_band_net"
hh = y3
for i in range(1, 5):
    s = V._cba(hh, sd, f"{p}.enc{i}.conv1", 1, 1, leaky=True)
    c = 8 << (i - 1)
    D = dm.engine.debug_fetch(f"vr.D{i-1}", (B, F_ >> (i - 1), W >> (i - 1), 3 * c))
    print(f"enc{i} skip", rel(D[..., 2 * c:], s.permute(0, 2, 3, 1).numpy()))
    hh = V._cba(s, sd, f"{p}.enc{i}.conv2", 2, 1, leaky=True)
    E = dm.engine.debug_fetch(f"vr.E{i-1}", (B, F_ >> i, W >> i, c))
    print(f"enc{i} out", rel(E, hh.permute(0, 2, 3, 1).numpy()))
a = p + ".aspp"
_, _, h_, w_ = hh.shape
pool = F.adaptive_avg_pool2d(hh, (1, None))
print("pool", rel(dm.engine.debug_fetch("vr.pool", (B, w_, 64)), pool[:, :, 0].permute(0, 2, 1).numpy()))
f1 = V._cba(pool, sd, a + ".conv1.1", 1, 0)
print("pool2", rel(dm.engine.debug_fetch("vr.pool2", (B, w_, 64)), f1[:, :, 0].permute(0, 2, 1).numpy()))
feats = [F.interpolate(f1, size=(h_, w_), mode="bilinear", align_corners=True), V._cba(hh, sd, a + ".conv2", 1, 0)]
feats += [V._sep(hh, sd, a + f".conv{j}", (4, 8, 16)[j - 3]) for j in (3, 4, 5)]
cat = dm.engine.debug_fetch("vr.cat", (B, h_, w_, 5 * 64))
for j, f in enumerate(feats):
    print("feat", j + 1, rel(cat[..., j * 64:(j + 1) * 64], f.permute(0, 2, 3, 1).numpy()))
bn = V._cba(torch.cat(feats, dim=1), sd, a + ".bottleneck.0", 1, 0)
print("bott", rel(dm.engine.debug_fetch("vr.bn", (B, h_, w_, 128)), bn.permute(0, 2, 3, 1).numpy()))
up = F.interpolate(bn, scale_factor=2, mode="bilinear", align_corners=True)
D3 = dm.engine.debug_fetch("vr.D3", (B, F_ >> 3, W >> 3, 3 * 64))
print("up4", rel(D3[..., :128], up.permute(0, 2, 3, 1).numpy()))
print("final", rel(y, g["hp_net_out"]))
