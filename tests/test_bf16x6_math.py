"""The arithmetic identity the bf16x6 row GEMM (csrc/kernels_gemm3.h) rests on, checked in numpy on the CPU:
a float32 is EXACTLY the sum of three bfloat16 numbers obtained by round-to-nearest-even residual splitting, and the six
products the kernel keeps reproduce a float32 product to 2^-23 relative -- the rounding an fp32 FMA chain commits anyway."""
import numpy as np


def bf16_rne(x):
    """float32 array -> the nearest bfloat16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 + a 16-bit shift give)"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)
    m = bf16_rne(r)
    r2 = (r - m).astype(np.float32)
    lo = bf16_rne(r2)
    return h, m, lo


def _samples(rng, n):
    mant = rng.uniform(1.0, 2.0, n)
    expo = rng.integers(-60, 60, n)
    sign = rng.choice([-1.0, 1.0], n)
    return (sign * mant * np.exp2(expo)).astype(np.float32)


def test_three_way_split_is_exact():
    rng = np.random.default_rng(0)
    x = np.concatenate([_samples(rng, 200000), rng.standard_normal(200000).astype(np.float32),
                        np.array([0.0, 1.0, -1.0, 3.0, 1 + 2.0 ** -23, 1 - 2.0 ** -24, 255.99998, 2.0 ** -100], np.float32)])
    h, m, lo = split3(x)
    # the residual subtractions are exact in float32 and the last residual is itself a bfloat16
    assert np.array_equal((x.astype(np.float64) - h.astype(np.float64)).astype(np.float32).astype(np.float64),
                          x.astype(np.float64) - h.astype(np.float64))
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    for part in (h, m, lo):                              # every part has at most 8 significand bits
        assert np.array_equal(bf16_rne(part), part)
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8) and np.all(np.abs(lo[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_six_products_match_a_float32_product():
    rng = np.random.default_rng(1)
    a, b = _samples(rng, 300000), rng.standard_normal(300000).astype(np.float32)
    ah, am, al = [v.astype(np.float64) for v in split3(a)]
    bh, bm, bl = [v.astype(np.float64) for v in split3(b)]
    kept = ah * bh + (ah * bm + am * bh) + (am * bm + ah * bl + al * bh)
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(kept - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -23, rel.max()           # dropped: am bl + al bm + al bl <= (2 * 2^-24 + 2^-32) |a b|
    assert np.sqrt(np.mean(rel ** 2)) < 2.0 ** -26


def test_gemm_emulation_is_fp32_grade():
    """K = 3072 dot products: six bf16 part-products accumulated in float32 against a plain float32 accumulation"""
    rng = np.random.default_rng(2)
    K, R = 3072, 64
    x = (3.0 * rng.standard_normal((R, K))).astype(np.float32)
    w = (rng.standard_normal((K, 48)) / np.sqrt(K)).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64)
    xs, ws = split3(x), split3(w)
    acc = np.zeros((R, 48), np.float32)
    for k0 in range(0, K, 32):                           # one MFMA k-step: the 32 products of a step summed, then added in fp32
        sl = slice(k0, k0 + 32)
        for i, j in ((0, 2), (2, 0), (1, 1), (1, 0), (0, 1), (0, 0)):
            acc += (xs[i][:, sl].astype(np.float64) @ ws[j][sl].astype(np.float64)).astype(np.float32)
    f32 = np.zeros((R, 48), np.float32)
    for k0 in range(0, K, 4):                            # the fp32 MFMA's k-step
        f32 += (x[:, k0:k0 + 4].astype(np.float64) @ w[k0:k0 + 4].astype(np.float64)).astype(np.float32)
    e6 = np.sqrt(np.mean((acc - ref) ** 2) / np.mean(ref ** 2))
    e32 = np.sqrt(np.mean((f32 - ref) ** 2) / np.mean(ref ** 2))
    assert e6 < 1e-6 and e6 < 1.5 * e32, (e6, e32)
