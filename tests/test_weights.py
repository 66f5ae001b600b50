"""BatchNorm folding (host logic) against the oracle's unfused forward."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import mdx_oracle as O
from audio_separator_amd.weights import fold_convtdf_state


def test_fold_matches_unfused():
    d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4, bias=True)
    sd = O.make_convtdf_state(d, seed=7)
    t = fold_convtdf_state(sd, d.num_blocks, d.l, tdf_bias=True)
    rng = np.random.default_rng(0)
    # TFC conv + BN + ReLU
    x = torch.tensor(rng.standard_normal((1, 8, 16, 32)).astype(np.float32))
    ref = F.relu(O._bn(F.conv2d(x, sd["encoding_blocks.0.tfc.H.0.0.weight"], sd["encoding_blocks.0.tfc.H.0.0.bias"], padding=1),
                       sd, "encoding_blocks.0.tfc.H.0.1"))
    got = F.relu(F.conv2d(x, torch.tensor(t["enc0.tfc0.w"]), torch.tensor(t["enc0.tfc0.b"]), padding=1))
    assert torch.allclose(ref, got, atol=2e-6, rtol=1e-5)
    # TDF linear + BN(c) + ReLU
    lin = F.linear(x, sd["encoding_blocks.0.tdf.0.weight"], sd["encoding_blocks.0.tdf.0.bias"])
    ref = F.relu(O._bn(lin, sd, "encoding_blocks.0.tdf.1"))
    sc = torch.tensor(t["enc0.tdf0.scale"])[None, :, None, None]
    sh = torch.tensor(t["enc0.tdf0.shift"])[None, :, None, None]
    got = F.relu(sc * (F.linear(x, torch.tensor(t["enc0.tdf0.w"])) + torch.tensor(t["enc0.tdf0.bias"])) + sh)
    assert torch.allclose(ref, got, atol=2e-6, rtol=1e-5)
    # transposed conv + BN
    xb = torch.tensor(rng.standard_normal((1, 24, 4, 8)).astype(np.float32))
    ref = O._bn(F.conv_transpose2d(xb, sd["us.0.0.weight"], sd["us.0.0.bias"], stride=2), sd, "us.0.1")
    got = F.conv_transpose2d(xb, torch.tensor(t["us0.w"]), torch.tensor(t["us0.b"]), stride=2)
    assert torch.allclose(ref, got, atol=2e-6, rtol=1e-5)


def test_names_cover_all_layers():
    d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4)
    t = fold_convtdf_state(O.make_convtdf_state(d, 1), d.num_blocks, d.l)
    for blk in ("enc0", "enc1", "mid", "dec0", "dec1"):
        for j in range(d.l):
            assert f"{blk}.tfc{j}.w" in t and f"{blk}.tfc{j}.b" in t
        for i in (0, 1):
            for s in ("w", "scale", "shift"):
                assert f"{blk}.tdf{i}.{s}" in t
    for k in ("first.w", "first.b", "final.w", "final.b", "ds0.w", "ds1.b", "us0.w", "us1.b"):
        assert k in t
    assert t["first.w"].shape == (8, 4) and t["final.w"].shape == (4, 8)
    assert t["us0.w"].shape == (24, 16, 2, 2) and t["ds0.w"].shape == (16, 8, 2, 2)
