"""The arithmetic the fp16 x 3 row GEMM (csrc/kernels_gemm3.h, template parameter H) rests on, restated in numpy on the CPU:
the two-part fp16 split of a block-scaled float32, what it loses where, the three products the kernel keeps, and the running
per-row exponent with its accumulator rescale -- stage by stage as the kernel does it (32 k per stage, fp32 accumulation)."""
import numpy as np


def scale_exp(m):
    """f16_scale_exp: e with m 2^e in [2^14, 2^15) (m > 0 finite); 15 for m == 0 (v_frexp_exp_i32_f32 returns 0 there)"""
    m = np.asarray(m, np.float32)
    _, ex = np.frexp(m)
    return np.where(m > 0, 15 - ex, 15).astype(np.int64)


def split2(x, e):
    """v 2^e = h + l with h = RNE_f16(v 2^e), l = RNE_f16(v 2^e - h); parts returned as float64 (fp16 values are exact in it)"""
    xs = np.ldexp(np.asarray(x, np.float32), e).astype(np.float32)
    with np.errstate(over="ignore"):
        h = xs.astype(np.float16)
        lo = (xs - h.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h.astype(np.float64), lo.astype(np.float64), xs.astype(np.float64)


def test_two_part_split_error_and_range():
    rng = np.random.default_rng(0)
    # a block whose largest element sits anywhere in [2^12, 2^15) after scaling (the kernel keeps two bits of headroom)
    for top in (2.0 ** 12, 2.0 ** 13.5, 2.0 ** 14.99):
        rel = np.exp2(-rng.uniform(0, 30, 200000))            # elements down to 2^-30 of the block maximum
        x = (rng.choice([-1.0, 1.0], rel.size) * top * rel).astype(np.float32)
        h, lo, xs = split2(x, 0)
        assert np.isfinite(h).all() and np.isfinite(lo).all()
        assert np.array_equal(xs - h, (xs - h).astype(np.float32).astype(np.float64)), "the residual subtraction is exact in fp32"
        err = np.abs(xs - h - lo)
        big = np.abs(xs) >= top * 2.0 ** -14                  # l stays above fp16's subnormal spacing
        assert np.all(err[big] <= np.abs(xs[big]) * 2.0 ** -23), (err[big] / np.abs(xs[big])).max()
        assert np.all(err <= 2.0 ** -25 + np.abs(xs) * 2.0 ** -23)   # everywhere else: half the subnormal spacing, absolute
        assert err.max() <= top * 2.0 ** -37 + top * 2.0 ** -23
    # nothing overflows while the block maximum is below 2^15 (fp16 max 65504)
    h, lo, _ = split2(np.float32(2.0 ** 15 * (1 - 2.0 ** -12)), 0)
    assert np.isfinite(h) and np.isfinite(lo)


def test_three_products_match_a_float32_product():
    rng = np.random.default_rng(1)
    a = (rng.uniform(1, 2, 300000) * np.exp2(rng.integers(0, 14, 300000)) * rng.choice([-1.0, 1.0], 300000)).astype(np.float32)
    b = (rng.uniform(1, 2, 300000) * np.exp2(rng.integers(0, 14, 300000))).astype(np.float32)
    ah, al, _ = split2(a, 0)
    bh, bl, _ = split2(b, 0)
    kept = ah * bh + (ah * bl + al * bh)
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(kept - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -21, rel.max()               # dropped al bl <= 2^-22, plus 2^-23 per operand
    assert np.sqrt(np.mean(rel ** 2)) < 2.0 ** -23


RISE, RISE_CAP = 10, 40          # kernels_gemm3.h F16X3_RISE / F16X3_RISE_CAP


def emulate_rows(x, w, stage=32, rise=True):
    """y = x @ w.T the way tdf3_kernel<H> does it: per-row running exponent (drops to need - 2 when a stage would pass 2^15; since round 6 rises
    to need - 2 when a stage's largest element is more than RISE bits below the range, at most RISE_CAP bits above the row's lowest exponent so
    far; accumulators multiplied by the exact power of two either way), one weight exponent per four output columns, fp32 accumulation of the
    three products.  rise = False: round 5's policy (drops only)."""
    x, w = np.asarray(x, np.float32), np.asarray(w, np.float32)
    M, K = x.shape
    N = w.shape[0]
    ew = np.zeros(N, np.int64)
    for t in range(0, N, 4):                            # one exponent per four output columns (w3h_split_kernel)
        m = np.abs(w[t:t + 4]).max()
        ew[t:t + 4] = scale_exp(m)[()] if m > 0 else 0
    wh, wl, _ = split2(w, ew[:, None])
    acc = np.zeros((M, N), np.float32)
    e_row = np.full(M, 200, np.int64)
    e_lo = np.full(M, 200, np.int64)
    drops = 0
    for k0 in range(0, K, stage):
        xs = x[:, k0:k0 + stage]
        top = np.abs(xs).max(axis=1)
        need = scale_exp(top)
        e_new = np.where(need < e_row, need - 2, e_row)
        if rise:
            up = (need > e_row + RISE) & (top > 0)
            e_new = np.where(up, np.minimum(need - 2, e_lo + RISE_CAP), e_new)
        e_lo = np.minimum(e_lo, e_new)
        de = e_new - e_row
        drops += int(np.count_nonzero(de[e_row != 200]))
        with np.errstate(under="ignore"):
            acc = (acc * np.ldexp(np.float32(1.0), de)[:, None].astype(np.float32)).astype(np.float32)   # 2^de, 0 below 2^-149
        e_row = e_new
        xh, xl, _ = split2(xs, e_row[:, None])
        assert np.isfinite(xh).all(), "a scaled element overflowed fp16"
        whk, wlk = wh[:, k0:k0 + stage], wl[:, k0:k0 + stage]
        for a_, b_ in ((xh, wlk), (xl, whk), (xh, whk)):                       # smallest terms first, one fp32 rounding per MFMA
            acc = (acc.astype(np.float64) + a_ @ b_.T).astype(np.float32)
    return np.ldexp(acc.astype(np.float64), -(e_row[:, None] + ew[None, :])), drops


def _rel(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2) / np.mean(b ** 2)))


def test_running_exponent_gemm_is_fp32_grade():
    rng = np.random.default_rng(2)
    M, K, N = 48, 1024, 40
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    w[16:32] *= 1e-6                                                            # a quiet weight tile
    w[32:36] *= 2.0 ** 10                                                       # a loud column group inside a tile
    cases = {
        "plain": rng.standard_normal((M, K)),
        "rows 2^40 apart": rng.standard_normal((M, K)) * np.exp2(rng.integers(-20, 21, (M, 1))),
        "growing along k": rng.standard_normal((M, K)) * np.exp2(24.0 * np.arange(K) / K),
        "decaying along k": rng.standard_normal((M, K)) * np.exp2(-20.0 * np.arange(K) / K),
        "zero first half": np.concatenate([np.zeros((M, K // 2)), rng.standard_normal((M, K // 2))], axis=1),
        "tiny": rng.standard_normal((M, K)) * 1e-35,
        "huge": rng.standard_normal((M, K)) * 1e33,
    }
    for name, x in cases.items():
        x = x.astype(np.float32)
        y, drops = emulate_rows(x, w)
        ref = x.astype(np.float64) @ w.astype(np.float64).T
        acc32 = np.zeros((M, N), np.float32)                                   # a plain fp32 accumulation in 32-wide steps for scale
        for k0 in range(0, K, 32):
            acc32 = (acc32.astype(np.float64) + x[:, k0:k0 + 32].astype(np.float64) @ w[:, k0:k0 + 32].astype(np.float64).T).astype(np.float32)
        e, e32 = _rel(y, ref), _rel(acc32.astype(np.float64), ref)
        rows = np.sqrt(((y - ref) ** 2).mean(axis=1) / (ref ** 2).mean(axis=1))
        assert e < 4 * e32 + 1e-7, (name, e, e32)
        assert rows.max() < 2e-6, (name, rows.max())                            # every row at its own scale
        if name == "growing along k":
            assert drops > M, "the stress case is meant to rescale accumulators after the first stage"
        if name == "plain":
            assert drops <= M // 4, (name, drops)                               # the maximum is met early: rescales are rare
        if name == "decaying along k":
            assert drops <= 2 * M + M // 4, (name, drops)                       # 2^-20 along k: the exponent follows in two rises of >= 10 bits


def test_running_exponent_follows_a_decay():
    """What the rise is for (round 6, ADVICE r5 on block exponents): x decays by 2^-bits along k and half of the output columns look at the quiet
    second half of k only.  Under round 5's drop-only policy the quiet half is carried at the loud half's exponent and those columns lose
    bits / 2 - 12 bits; with the rise every element stays within RISE + 3 bits of its stage's range -- 22-bit products throughout, fp32-grade
    at any spread.  (The WEIGHTS' exponent is per four output columns over all of k, chosen when the image is built: a weight row that spans
    2^30 along k is outside the contract of the arithmetic, and of no net this library loads.)"""
    rng = np.random.default_rng(12)
    M, K, N = 16, 2048, 24
    for bits in (24, 48, 90):
        dec = np.exp2(-bits * np.arange(K) / (K - 1))
        x = (rng.standard_normal((M, K)) * dec).astype(np.float32)
        w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        w[: N // 2, : K // 2] = 0                                               # columns 0 .. N/2 - 1: the quiet half of k only
        ref = x.astype(np.float64) @ w.astype(np.float64).T
        y_new, n_new = emulate_rows(x, w, rise=True)
        y_old, _ = emulate_rows(x, w, rise=False)
        q = slice(0, N // 2)
        e_new, e_old, e_loud = _rel(y_new[:, q], ref[:, q]), _rel(y_old[:, q], ref[:, q]), _rel(y_new[:, N // 2:], ref[:, N // 2:])
        assert e_new < 4e-7 and e_loud < 4e-7, (bits, e_new, e_loud)
        print(f"decay 2^-{bits}: quiet columns rel-RMS {e_new:.2e} (rise) / {e_old:.2e} (drop only), loud columns {e_loud:.2e}, {n_new} rescales")
        if bits >= 48:
            assert e_old > 10 * e_new, (bits, e_old, e_new)                     # the case was real: drop-only loses the quiet half
        assert n_new <= (bits // RISE + 1) * M, (bits, n_new)                   # a handful of rescales per row


def test_nonfinite_rows_stay_local():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((8, 128)).astype(np.float32)
    w = (rng.standard_normal((16, 128)) / 11.0).astype(np.float32)
    clean, _ = emulate_rows(x, w)
    x2 = x.copy()
    x2[2, 70] = np.inf
    x2[5, 3] = np.nan
    with np.errstate(invalid="ignore", over="ignore"):
        need = scale_exp(np.where(np.isfinite(x2), np.abs(x2), 0).max(axis=1))  # v_max skips NaN; an Inf row's exponent is its own business
    assert np.array_equal(need[[0, 1, 3, 4, 6, 7]], scale_exp(np.abs(x).max(axis=1))[[0, 1, 3, 4, 6, 7]])
    # rows 2 and 5 are the only ones whose operands change: per-row exponents, per-row accumulators
    keep = [0, 1, 3, 4, 6, 7]
    again, _ = emulate_rows(x[keep], w)
    assert np.array_equal(again, clean[keep])
