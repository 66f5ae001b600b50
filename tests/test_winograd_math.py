"""The Winograd F(2x2, 3x3) identities the 3x3 TFC kernel is built on (csrc/kernels_wino.h), restated in numpy and checked against a
direct convolution: Y = A^T [ sum_c (G g G^T) * (B^T d B) ] A on every 2x2 output tile (Lavin & Gray 2016).  The kernel's host side
computes U = G g G^T in float64 (asx.hip: conv_pack) and its device side forms B^T d B and A^T m A with the additions written out
below; this test pins the matrices and the tile / halo indexing (pad 1, tiles starting at even pixels), not the HIP code -- the
GPU parity of the kernel itself is tests/test_gpu_parity.py::test_conv3x3_winograd."""
import numpy as np

G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float64)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def direct(x, w):
    cin, T, F = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    y = np.zeros((w.shape[0], T, F))
    for dy in range(3):
        for dx in range(3):
            y += np.einsum("oc,ctf->otf", w[:, :, dy, dx], xp[:, dy:dy + T, dx:dx + F])
    return y


def winograd(x, w):
    cin, T, F = x.shape
    assert T % 2 == 0 and F % 2 == 0
    U = np.einsum("ai,ocij,bj->ocab", G, w, G)                     # G g G^T per (cout, cin)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    y = np.zeros((w.shape[0], T, F))
    for ty in range(T // 2):
        for tx in range(F // 2):
            d = xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]        # the 4 x 4 patch of output tile (ty, tx), halo included
            V = np.einsum("ai,cij,bj->cab", BT, d, BT)             # B^T d B per channel
            M = np.einsum("ocab,cab->oab", U, V)                   # sixteen element-wise products, summed over channels
            y[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("pa,oab,qb->opq", AT, M, AT)
    return y


def test_winograd_equals_direct_convolution():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((7, 6, 10))
    w = rng.standard_normal((5, 7, 3, 3))
    assert np.allclose(winograd(x, w), direct(x, w), rtol=0, atol=1e-12)


def test_row_and_column_transforms_are_the_kernels_additions():
    # the device code writes the transforms as additions: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3), the same on columns;
    # outputs (m0 + m1 + m2, m1 - m2 - m3)
    d = np.random.default_rng(6).standard_normal((4, 4))
    r = np.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]])
    v = np.stack([r[:, 0] - r[:, 2], r[:, 1] + r[:, 2], r[:, 2] - r[:, 1], r[:, 1] - r[:, 3]], axis=1)
    assert np.allclose(v, BT @ d @ BT.T, atol=1e-15)
    m = np.random.default_rng(7).standard_normal((4, 4))
    c = np.stack([m[0] + m[1] + m[2], m[1] - m[2] - m[3]])
    y = np.stack([c[:, 0] + c[:, 1] + c[:, 2], c[:, 1] - c[:, 2] - c[:, 3]], axis=1)
    assert np.allclose(y, AT @ m @ AT.T, atol=1e-15)


def test_multiply_count():
    # 16 multiply-adds per 2 x 2 output tile and channel pair against 36 for the direct form: the 4/9 of bench.py's `executed` rate
    assert (4 * 4) / (2 * 2 * 9) == 4 / 9
