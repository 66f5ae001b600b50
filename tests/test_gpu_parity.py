"""GPU parity tests: the HIP path (through the C ABI, libasx.so) against the CPU
oracle and the committed golden vectors.  Floating-point bar: relative RMS
<= 1e-4 of the reference signal (north star: stems within 1e-4 RMS, fp32);
individual stages are held to tighter bounds stated per test.
"""
import os

import numpy as np
import pytest

from oracle import mdx_oracle as O

pytestmark = pytest.mark.gpu

TOL_STEM = 1e-4


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


@pytest.fixture(scope="module")
def A():
    import audio_separator_amd as A
    return A


def small_cfg(A, overlap=0.25, denoise=False, max_batch=0):
    return A.MDXConfig(n_fft=96, hop_length=16, dim_f=32, segment_size=16, overlap=overlap, enable_denoise=denoise,
                       max_batch=max_batch)


SMALL_DIMS = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4, bias=False)


def small_engine(A, overlap=0.25, denoise=False, max_batch=0, seed=3, bias=False):
    d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4, bias=bias)
    sd = O.make_convtdf_state(d, seed=seed)
    eng = A.Engine(small_cfg(A, overlap, denoise, max_batch))
    eng.load_net(A.NetConfig(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4, tdf_bias=bias),
                 A.fold_convtdf_state(sd, d.num_blocks, d.l, tdf_bias=bias))
    return eng, sd, d


# ---------------------------------------------------------------------------
# STFT / iSTFT
# ---------------------------------------------------------------------------
def test_stft_small_golden(A, golden_dir):
    g = np.load(os.path.join(golden_dir, "stft_small.npz"))
    eng = A.Engine(A.MDXConfig(n_fft=96, hop_length=16, dim_f=40, segment_size=16))
    x = (0.5 * np.random.default_rng(int(g["x_seed"])).standard_normal((2, 2, 240))).astype(np.float32)
    X = eng.stft(x)
    assert X.shape == g["X"].shape
    assert rel_rms(X, g["X"]) < 5e-6, rel_rms(X, g["X"])
    S = np.random.default_rng(int(g["s_seed"])).standard_normal(X.shape).astype(np.float32)
    y = eng.istft(S)
    assert rel_rms(y, g["y"]) < 5e-6, rel_rms(y, g["y"])


def test_stft_hq3_golden_and_oracle(A, golden_dir):
    g = np.load(os.path.join(golden_dir, "stft_hq3_subsample.npz"))
    eng = A.Engine(A.MDXConfig())
    C = 1024 * 255
    x = (0.3 * np.random.default_rng(int(g["x_seed"])).standard_normal((1, 2, C))).astype(np.float32)
    X = eng.stft(x)
    assert X.shape == (1, 4, 3072, 256)
    assert rel_rms(X[:, :, g["fsel"]][:, :, :, g["tsel"]], g["X_sub"]) < 5e-6
    assert abs(np.abs(X.astype(np.float64)).sum() / float(g["X_abs_sum"]) - 1) < 1e-5
    Xo = O.stft_forward(x, 6144, 1024, 3072)
    assert rel_rms(X, Xo) < 5e-6, rel_rms(X, Xo)
    S = np.random.default_rng(int(g["s_seed"])).standard_normal((1, 4, 3072, 256)).astype(np.float32)
    y = eng.istft(S)
    assert rel_rms(y[:, :, g["ysel"]], g["y_sub"]) < 5e-6
    yo = O.stft_inverse(S, 6144, 1024)
    assert rel_rms(y, yo) < 5e-6, rel_rms(y, yo)


@pytest.mark.parametrize("n_fft,hop,dim_f,T", [(2048, 512, 1025, 32), (4096, 1024, 2048, 20), (5120, 1024, 2560, 12),
                                               (7680, 1024, 3072, 18), (160, 32, 81, 9)])
def test_stft_other_radices(A, n_fft, hop, dim_f, T):
    # 2048 = pure radix-4/2, 5120/7680/160 exercise radix 5; dim_f = n_fft/2+1 includes the Nyquist bin
    st = A.STFT(None, n_fft, hop, dim_f, 0)
    x = np.random.default_rng(n_fft).standard_normal((2, 2, hop * (T - 1))).astype(np.float32)
    X = st(x)
    Xo = O.stft_forward(x, n_fft, hop, dim_f)
    assert X.shape == Xo.shape
    assert rel_rms(X, Xo) < 5e-6, rel_rms(X, Xo)
    S = np.random.default_rng(n_fft + 1).standard_normal(Xo.shape).astype(np.float32)
    y = st.inverse(S)
    yo = O.stft_inverse(S, n_fft, hop)
    assert rel_rms(y, yo) < 5e-6, rel_rms(y, yo)


def test_stft_shape_contract_torch(A):
    # the reference's own unit contract (tests/unit/test_stft.py:43-75,124-138), torch tensors in and out
    import torch
    st = A.STFT(None, 2048, 512, 1025, torch.device("cuda:0"))
    out = st(torch.rand(1, 2, 16000))
    assert isinstance(out, torch.Tensor) and tuple(out.shape[-2:]) == (1025, 16000 // 512 + 1)
    inv = st.inverse(torch.rand(1, 4, 1025, 32))
    assert tuple(inv.shape) == (1, 2, 15872)


def test_stft_roundtrip_full_chunk(A):
    # size-independent property: istft(stft(x)) == x when no bin is dropped (dim_f = n/2+1)
    eng = A.Engine(A.MDXConfig(n_fft=6144, hop_length=1024, dim_f=3073, segment_size=256))
    x = np.random.default_rng(5).standard_normal((2, 2, 1024 * 255)).astype(np.float32)
    y = eng.istft(eng.stft(x))
    assert rel_rms(y, x) < 2e-6, rel_rms(y, x)


# ---------------------------------------------------------------------------
# single layers
# ---------------------------------------------------------------------------
def _torch_ref(op, x, w, b, aux=None, relu=True):
    import torch
    import torch.nn.functional as F
    xt, wt, bt = torch.tensor(x), torch.tensor(w), torch.tensor(b)
    if op == "conv3x3":
        y = F.conv2d(xt, wt, bt, padding=1)
        if relu:
            y = F.relu(y)
    elif op == "down":
        y = F.relu(F.conv2d(xt, wt, bt, stride=2))
    elif op == "conv1x1":
        y = F.conv2d(xt, wt[:, :, None, None], bt)
        if relu:
            y = F.relu(y)
    else:
        y = F.relu(F.conv_transpose2d(xt, wt, bt, stride=2)) * torch.tensor(aux)
    return y.numpy()


CONV_CASES = [
    # op, B, cin, cout, T, F
    ("conv3x3", 1, 48, 48, 16, 128), ("conv3x3", 2, 96, 96, 8, 64), ("conv3x3", 1, 32, 32, 24, 200),
    ("conv3x3", 2, 8, 8, 16, 32), ("conv3x3", 1, 24, 24, 4, 8), ("conv3x3", 1, 16, 16, 5, 7),
    ("conv3x3", 1, 144, 144, 8, 96), ("conv3x3", 1, 80, 80, 8, 64),
    ("down", 1, 48, 96, 16, 128), ("down", 2, 8, 16, 16, 32), ("down", 1, 96, 144, 8, 256), ("down", 1, 16, 24, 8, 16),
    ("down", 1, 48, 96, 8, 30), ("down", 1, 50, 144, 6, 72),   # two-channel stages: register-staged loader (F % 4 != 0), cin % 4 != 0
    ("up", 1, 96, 48, 8, 64), ("up", 2, 16, 8, 8, 16), ("up", 1, 144, 96, 4, 96), ("up", 1, 24, 16, 4, 8),
    ("up", 1, 64, 32, 6, 40),
    ("conv1x1", 2, 4, 48, 16, 128), ("conv1x1", 1, 48, 4, 16, 128), ("conv1x1", 2, 4, 8, 16, 32),
    ("conv1x1", 1, 8, 4, 16, 32), ("conv1x1", 1, 32, 4, 8, 64),
]


@pytest.mark.parametrize("op,B,cin,cout,T,F", CONV_CASES)
def test_conv_layers(A, op, B, cin, cout, T, F):
    eng = A.Engine(small_cfg(A))
    rng = np.random.default_rng(cin * 1000 + cout + T)
    x = rng.standard_normal((B, cin, T, F)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    aux = None
    relu = op != "conv1x1" or cout != 4
    if op == "conv3x3":
        w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    elif op == "down":
        w = (rng.standard_normal((cout, cin, 2, 2)) / np.sqrt(4 * cin)).astype(np.float32)
    elif op == "conv1x1":
        w = (rng.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32)
    else:
        w = (rng.standard_normal((cin, cout, 2, 2)) / np.sqrt(cin)).astype(np.float32)
        aux = rng.standard_normal((B, cout, 2 * T, 2 * F)).astype(np.float32)
    y = eng.op_conv(op, x, w, b, aux=aux, relu=relu)
    ref = _torch_ref(op, x, w, b, aux, relu)
    assert y.shape == ref.shape
    assert np.isfinite(y).all(), "unwritten (NaN canary) output elements"
    assert max_abs(y, ref) < 2e-5, (max_abs(y, ref), rel_rms(y, ref))


@pytest.mark.parametrize("B,cin,cout,T,F", [(1, 48, 96, 16, 128), (2, 96, 144, 8, 256), (1, 144, 192, 12, 136), (2, 8, 16, 16, 32), (1, 50, 144, 6, 72),
                                             (1, 20, 40, 7, 132), (1, 240, 288, 4, 192), (3, 48, 96, 2, 8)])
def test_down_conv_bf16x6(A, B, cin, cout, T, F):
    """conv_down6_kernel (csrc/kernels_updown6.h, round 6): the 2 x 2 / stride-2 conv between the levels on the 16-bit matrix pipe -- six bf16 products on
    exactly split operands, fp32 accumulation.  Against float64 and against the fp32-MFMA kernel it replaces (option conv_down_bf16x6 = 0), with proof
    of which ran; channel counts off the 8 / 48 grids, odd output heights, output widths off the 64-pixel tile, several batch items."""
    import torch
    eng = A.Engine(small_cfg(A))
    assert eng.option("conv_down_bf16x6") == 1 and eng.option("gemm_bf16x6") == 1
    rng = np.random.default_rng(cin * 100 + cout + F)
    x = (rng.standard_normal((B, cin, T, F)) * np.exp2(rng.integers(-6, 7, (B, cin, 1, 1)))).astype(np.float32)   # channels 2^12 apart
    w = (rng.standard_normal((cout, cin, 2, 2)) / np.sqrt(4 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    n0 = eng.counter("down6_launches")
    y = eng.op_conv("down", x, w, b, relu=True)
    assert eng.counter("down6_launches") == n0 + (1 if F % 8 == 0 else 0), "conv_down6_kernel did not run"   # (16-byte output rows: else the fp32 kernel)
    assert np.array_equal(y, eng.op_conv("down", x, w, b, relu=True)), "not deterministic"
    eng.set_option("conv_down_bf16x6", 0)
    n0 = eng.counter("down6_launches")
    y32 = eng.op_conv("down", x, w, b, relu=True)
    assert eng.counter("down6_launches") == n0, "the fp32 run went through conv_down6_kernel"
    assert np.isfinite(y).all(), "unwritten (NaN canary) output elements"
    r64 = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=2)).numpy()
    mag = torch.nn.functional.conv2d(torch.from_numpy(x).double().abs(), torch.from_numpy(w).double().abs(), stride=2).numpy() + np.abs(b)[None, :, None, None]
    d6, d32 = np.abs(y - r64), np.abs(y32 - r64)
    assert (d6 <= 8e-7 * mag + 1e-30).all(), float((d6 / mag).max())           # every output against the size of its own products (K <= 960: a few fp32 roundings of the running sum)
    e6, e32 = rel_rms(y, r64), rel_rms(y32, r64)
    print(f"down conv {cin} -> {cout}: rel-RMS vs float64 {e6:.2e} (bf16 x 6) / {e32:.2e} (fp32 MFMA)")
    assert e6 <= 1.25 * e32 + 1e-8, (e6, e32)                                # at least as close as the fp32 kernel


@pytest.mark.parametrize("B,cin,cout,T,F", [(1, 96, 48, 8, 64), (2, 144, 96, 4, 96), (1, 192, 144, 6, 72), (2, 16, 8, 8, 16), (1, 24, 16, 5, 8),
                                             (1, 64, 32, 6, 40), (1, 288, 240, 2, 96), (1, 100, 20, 3, 68)])
def test_up_conv_bf16x6(A, B, cin, cout, T, F):
    """conv_up6_kernel (csrc/kernels_updown6.h, round 6): the transposed 2 x 2 / stride-2 conv of the decoder with its `x *= skip`, on the 16-bit matrix pipe
    (six bf16 products on exactly split operands).  Against float64 and the fp32-MFMA kernel (option conv_up_bf16x6 = 0), with proof of which ran; every
    virtual-tile grouping (Cout = 48 / 96 -> 6, 32 -> 4, 8 / 16 / 20 -> 2), channel counts off the 32-channel stage, odd heights, widths off the tile."""
    import torch
    eng = A.Engine(small_cfg(A))
    assert eng.option("conv_up_bf16x6") == 1
    rng = np.random.default_rng(cin * 100 + cout + F)
    x = (rng.standard_normal((B, cin, T, F)) * np.exp2(rng.integers(-6, 7, (B, cin, 1, 1)))).astype(np.float32)
    w = (rng.standard_normal((cin, cout, 2, 2)) / np.sqrt(cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    skip = rng.standard_normal((B, cout, 2 * T, 2 * F)).astype(np.float32)
    n0 = eng.counter("up6_launches")
    y = eng.op_conv("up", x, w, b, aux=skip, relu=True)
    assert eng.counter("up6_launches") == n0 + 1, "conv_up6_kernel did not run"
    assert np.array_equal(y, eng.op_conv("up", x, w, b, aux=skip, relu=True)), "not deterministic"
    eng.set_option("conv_up_bf16x6", 0)
    n0 = eng.counter("up6_launches")
    y32 = eng.op_conv("up", x, w, b, aux=skip, relu=True)
    assert eng.counter("up6_launches") == n0, "the fp32 run went through conv_up6_kernel"
    assert np.isfinite(y).all(), "unwritten (NaN canary) output elements"
    xt, wt = torch.from_numpy(x).double(), torch.from_numpy(w).double()
    r64 = (torch.relu(torch.nn.functional.conv_transpose2d(xt, wt, torch.from_numpy(b).double(), stride=2)) * torch.from_numpy(skip).double()).numpy()
    mag = (torch.nn.functional.conv_transpose2d(xt.abs(), wt.abs(), stride=2).numpy() + np.abs(b)[None, :, None, None]) * np.abs(skip)
    d6 = np.abs(y - r64)
    assert (d6 <= 8e-7 * mag + 1e-30).all(), float((d6 / np.maximum(mag, 1e-300)).max())
    e6, e32 = rel_rms(y, r64), rel_rms(y32, r64)
    print(f"up conv {cin} -> {cout}: rel-RMS vs float64 {e6:.2e} (bf16 x 6) / {e32:.2e} (fp32 MFMA)")
    assert e6 <= 1.25 * e32 + 1e-8, (e6, e32)


TDF_CASES = [
    # B, c, T, K, N, bias, res
    (1, 48, 16, 3072, 384, False, False), (1, 48, 16, 384, 3072, False, True), (2, 8, 16, 32, 8, True, False),
    (2, 8, 16, 8, 32, True, True), (1, 24, 4, 8, 2, False, False), (1, 24, 4, 2, 8, False, True),
    (1, 96, 8, 1536, 192, False, False), (1, 144, 4, 96, 768, False, True), (1, 5, 3, 50, 70, True, True),
    (1, 288, 8, 96, 12, False, False), (1, 288, 8, 12, 96, False, True),
]


@pytest.mark.parametrize("B,c,T,K,N,bias,res", TDF_CASES)
def test_tdf_layers(A, B, c, T, K, N, bias, res):
    import torch
    import torch.nn.functional as F
    eng = A.Engine(small_cfg(A))
    rng = np.random.default_rng(K * 7 + N)
    x = rng.standard_normal((B, c, T, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bv = rng.standard_normal(N).astype(np.float32) if bias else None
    sc = (0.5 + rng.random(c)).astype(np.float32)
    sh = (0.2 * rng.standard_normal(c)).astype(np.float32)
    r = rng.standard_normal((B, c, T, N)).astype(np.float32) if res else None
    y = eng.op_tdf(x, w, bv, sc, sh, r)
    lin = F.linear(torch.tensor(x), torch.tensor(w), torch.tensor(bv) if bias else None).numpy()
    ref = np.maximum(sc[None, :, None, None] * lin + sh[None, :, None, None], 0)
    if res:
        ref = ref + r
    assert np.isfinite(y).all(), "unwritten (NaN canary) output elements"
    assert max_abs(y, ref) < 3e-5, (max_abs(y, ref), rel_rms(y, ref))


# Split-operand row GEMM (csrc/kernels_gemm3.h), both arithmetics -- bf16 x 6 (exact three-way split) and fp16 x 3 (block-scaled
# two-way split, the default since round 5): the same layer through the split-operand kernel and through the fp32-MFMA kernel, both
# against a float64 GEMM.  The bar is "at least as close as the fp32 kernel" (a reduced-precision shortcut would fail it by 100x),
# plus proof that the kernel under test is the one that ran (library launch counters).
ARITH = ["bf16x6", "f16x3"]


def _set_arith(eng, arith):
    eng.set_option("gemm_bf16x6", 1)
    eng.set_option("gemm_f16x3", 1 if arith == "f16x3" else 0)


def _ran(eng, arith, n0, h0, launches=1):
    """the split-operand kernel of `arith` ran `launches` times since the counters read (n0, h0)"""
    n, h = eng.counter("tdf3_launches") - n0, eng.counter("tdf3h_launches") - h0
    return n == launches and h == (launches if arith == "f16x3" else 0)


ROWGEMM_CASES = [
    # B, c, T, K, N, res, x scale
    (2, 48, 64, 3072, 384, False, 3.0), (2, 48, 64, 384, 3072, True, 1.0), (1, 96, 37, 1536, 192, False, 1e4),
    (1, 3, 41, 512, 2048, True, 1e-3), (1, 1, 1000, 64, 72, True, 1.0), (3, 5, 7, 128, 200, False, 30.0),
    (2, 144, 16, 96, 768, True, 1.0), (1, 2, 33, 160, 136, False, 5.0),      # odd stage counts (K = 96, 160)
]


@pytest.mark.parametrize("arith", ARITH)
@pytest.mark.parametrize("B,c,T,K,N,res,xs", ROWGEMM_CASES)
def test_rowgemm_bf16x6_vs_float64(A, B, c, T, K, N, res, xs, arith):
    eng = A.Engine(small_cfg(A))
    rng = np.random.default_rng(K * 11 + N)
    x = (xs * rng.standard_normal((B, c, T, K)) * np.exp2(rng.integers(-6, 7, (B, c, T, 1)))).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    sc = (0.5 + rng.random(c)).astype(np.float32)
    sh = (0.2 * xs * rng.standard_normal(c)).astype(np.float32)
    r = (xs * rng.standard_normal((B, c, T, N))).astype(np.float32) if res else None
    ref = np.maximum(sc[None, :, None, None].astype(np.float64) * (x.astype(np.float64) @ w.astype(np.float64).T)
                     + sh[None, :, None, None], 0)
    if res:
        ref = ref + r
    try:
        _set_arith(eng, arith)
        n0, h0 = eng.counter("tdf3_launches"), eng.counter("tdf3h_launches")
        y6 = eng.op_tdf(x, w, None, sc, sh, r)
        assert _ran(eng, arith, n0, h0), f"the {arith} kernel did not run"
        y6b = eng.op_tdf(x, w, None, sc, sh, r)
        eng.set_option("gemm_bf16x6", 0)
        n1 = eng.counter("tdf3_launches")
        y32 = eng.op_tdf(x, w, None, sc, sh, r)
        assert eng.counter("tdf3_launches") == n1, "the fp32 run went through the split-operand kernel"
    finally:
        eng.set_option("gemm_bf16x6", 1)
    assert np.isfinite(y6).all(), "unwritten (NaN canary) output elements"
    assert np.array_equal(y6, y6b), "not deterministic"
    e6, e32 = rel_rms(y6, ref), rel_rms(y32, ref)
    assert e6 < 2e-6 and e6 <= 1.25 * e32 + 1e-8, (e6, e32)
    assert np.abs(y6 - ref).max() <= 1.5 * np.abs(y32 - ref).max() + 1e-6 * np.abs(ref).max(), (np.abs(y6 - ref).max(), np.abs(y32 - ref).max())


# fp16 x 3 keeps one running exponent per ROW of x and one per four output columns of W.  What a block exponent has to survive: rows of
# one workgroup tile 2^18 apart, a 2^-20 decay along k (a spectrum: loud low bins, quiet high bins), output columns that look at the quiet
# half of k only, a weight tile 1e-6 of its neighbours, a maximum that keeps growing along k (accumulators rescaled stage after
# stage), all-zero stages and all-zero rows.  Bars: whole result as close to float64 as the fp32 kernel; EVERY row and EVERY column group
# within 4e-6 of its own scale (with ONE exponent per 128-row tile -- the first form of the kernel -- the "decay" case missed this by
# 1e-2 on its quiet rows: that is what the per-row exponents are for).
@pytest.mark.parametrize("shape", ["decay", "decay60", "growing", "zeros", "chan4"])
def test_rowgemm_f16x3_block_exponent(A, shape):
    # "decay60" (round 6): x falls by 2^-60 along k -- the running row exponent has to RISE after the loud stages or the columns that look at
    # the quiet half of k lose everything (tests/test_f16x3_math.py::test_running_exponent_follows_a_decay is the CPU model of the policy)
    eng = A.Engine(small_cfg(A))
    B, c, T, K, N = 1, 2, 192, 1024, 272
    rng = np.random.default_rng(91)
    x = rng.standard_normal((B, c, T, K))
    k = np.arange(K)
    if shape.startswith("decay"):
        x *= np.exp2((-60.0 if shape == "decay60" else -20.0) * k / K)[None, None, None, :] * np.exp2(3.0 * (np.arange(T) % 7) - 9)[None, None, :, None]
    elif shape == "growing":
        x *= np.exp2(24.0 * k / K)[None, None, None, :]
    else:
        x[..., : K // 2] = 0.0                         # the first stages are all zero, then ordinary data
        x[:, :, 100:, :] = 0.0                         # and a tile whose rows are zero throughout
    x = x.astype(np.float32)
    w = rng.standard_normal((N, K)) / np.sqrt(K)
    w[:16, : K // 2] = 0.0                             # these columns see the second half of k only
    w[16:32] *= 1e-6                                   # a quiet weight tile
    if shape == "chan4":                               # W's exponents are per FOUR output columns: groups 2^40 apart inside one 16-column tile
        w *= np.repeat(np.exp2(rng.integers(-20, 21, N // 4)), 4)[:, None]
    w = w.astype(np.float32)
    sc, sh = np.ones(c, np.float32), np.zeros(c, np.float32)
    lin = x.astype(np.float64) @ w.astype(np.float64).T
    ref = np.maximum(lin, 0)
    try:
        _set_arith(eng, "f16x3")
        n0, h0 = eng.counter("tdf3_launches"), eng.counter("tdf3h_launches")
        y = eng.op_tdf(x, w, None, sc, sh, None)
        assert _ran(eng, "f16x3", n0, h0)
        eng.set_option("gemm_bf16x6", 0)
        y32 = eng.op_tdf(x, w, None, sc, sh, None)
    finally:
        eng.set_option("gemm_bf16x6", 1)
    assert np.isfinite(y).all()
    e, e32 = rel_rms(y, ref), rel_rms(y32, ref)
    assert e < 2e-6 and e <= 1.25 * e32 + 1e-8, (e, e32)
    worst = 0.0
    for cols in (slice(0, 16), slice(16, 32), slice(32, N)):
        yy, rr, ll = y[..., cols].reshape(-1, y[..., cols].shape[-1]), ref[..., cols].reshape(-1, ref[..., cols].shape[-1]), \
            lin[..., cols].reshape(-1, ref[..., cols].shape[-1])
        scale = np.sqrt((ll ** 2).mean(axis=1))        # the row's own scale in this column group (pre-ReLU)
        err = np.sqrt(((yy - rr) ** 2).mean(axis=1))
        ok = scale > 0
        worst = max(worst, float((err[ok] / scale[ok]).max()) if ok.any() else 0.0)
        assert (err[~ok] == 0).all(), "an all-zero row did not come out zero"
    if shape == "chan4":                               # ... and every COLUMN at its own scale
        cs = np.sqrt((lin.reshape(-1, N) ** 2).mean(axis=0))
        ce = np.sqrt(((y.reshape(-1, N) - ref.reshape(-1, N)) ** 2).mean(axis=0))
        worst = max(worst, float((ce / cs).max()))
    print(f"fp16 x 3 row GEMM, {shape}: rel-RMS {e:.3e} (fp32-MFMA {e32:.3e}), worst row x column-group error over its own scale {worst:.3e}")
    assert worst < 4e-6, worst


# ---- edge values of the exact three-way split ------------------------------------------------------------
# v = h + m + l with h = RNE_bf16(v): what happens where bf16 runs out of range.
#  * tiny operands: bfloat16 has float32's exponent range, so h and m stay normal down to ~1e-36, but l = v - h - m sits 2^-16 below v
#    and becomes a bf16 SUBNORMAL below |v| ~ 8e-34; whatever the matrix pipe does with subnormal operands, the dropped part is
#    <= 2^-16 |v| of an operand that is itself 1e-34 -- the test pins that the result stays within 1e-4 (north-star bar) of float64
#    there and keeps the full "as close as the fp32 kernel" bar at 1e-30;
#  * huge operands (1e35, products up to ~1e37: no overflow in either kernel): same bar as ordinary data;
#  * non-finite operands are OUTSIDE the parity domain (the reference turns the whole chunk into NaN: torch.relu keeps NaN, the next
#    layer spreads it; its VR path scrubs them first, vr_separator.py:181-182), but what the kernels do is pinned here so that a
#    change is noticed: h = Inf gives v - h = NaN, so an Inf / NaN in x makes every accumulator of its ROW NaN in the bf16 x 6
#    kernel where the fp32-MFMA kernel holds +-Inf (NaN when signs cancel); the TDF epilogue's ReLU is a maxNum (v_max_f32), which
#    returns 0 for NaN -- so the affected row comes out 0 here and {Inf, 0} there.  No other row may change by a single bit.
# Pair images (round 6; kernels_gemm3.h "operands split by their producer", option "gemm_pair_images" -- EXPERIMENTAL builds): the bottleneck activations of a TDF block
# written by the first linear's epilogue as the two fp16 parts the second linear multiplies.  Against the same block through fp32 activations and
# against float64; the decoded image against the fp32 activations; proof by counter that the pair-image reader ran.
PAIR_CASES = [
    # B, c, T, F, F/8, what
    (2, 48, 228, 3072, 384, "plain"),       # level 0 of HQ_3, enough rows (171 tiles) for the launcher's 128 x 192 tile: two 192-column spans
    (2, 48, 64, 3072, 384, "plain"),        # the same layer on few rows: 128-column tiles, three spans
    (1, 96, 40, 1536, 192, "plain"),        # level 1
    (1, 3, 37, 768, 96, "plain"),           # 64 x 128 tile form, ragged rows, one span
    (2, 5, 33, 1024, 224, "plain"),         # last span narrower than a tile, ragged rows
    (1, 4, 64, 3072, 384, "quiet_span"),    # the first span 2^-12 of the second: each keeps its own exponent
    (1, 4, 64, 3072, 384, "loud_rows"),     # rows of a tile 2^16 apart
    (1, 4, 64, 3072, 384, "dead_rows"),     # ReLU leaves whole rows of zeros (exponent of an all-zero span)
]


@pytest.mark.parametrize("B,c,T,F,F8,what", PAIR_CASES)
def test_tdf_block_pair_image(A, B, c, T, F, F8, what):
    rng = np.random.default_rng(F + F8 + len(what))
    x = rng.standard_normal((B, c, T, F)).astype(np.float32)
    w0 = (rng.standard_normal((F8, F)) / np.sqrt(F)).astype(np.float32)
    w1 = (rng.standard_normal((F, F8)) / np.sqrt(F8)).astype(np.float32)
    sc0, sc1 = (0.5 + rng.random(c)).astype(np.float32), (0.5 + rng.random(c)).astype(np.float32)
    sh0, sh1 = (0.05 * rng.standard_normal(c)).astype(np.float32), (0.2 * rng.standard_normal(c)).astype(np.float32)
    if what == "quiet_span":
        w0[: F8 // 2] *= 2.0 ** -12
        sh0[:] = 0
    elif what == "loud_rows":
        x *= (2.0 ** (4 * (np.arange(T) % 5) - 8)).astype(np.float32)[None, None, :, None]
    elif what == "dead_rows":
        sh0[:] = 0
        x[:, :, ::3, :] = 0                                  # every third row: h = relu(0) = 0 in every column
    eng = A.Engine(small_cfg(A))
    _set_or_skip(eng, "gemm_pair_images", 1)               # experimental builds only (measured round 6: no gain in the nets, profiles/NOTES.md)
    assert eng.option("gemm_f16x3") == 1
    n0, p0 = eng.counter("tdf3h_launches"), eng.counter("tdf3_pair_image_launches")
    y1, h1 = eng.op_tdf_block(x, w0, sc0, sh0, w1, sc1, sh1, return_hidden=True)
    assert eng.counter("tdf3h_launches") - n0 == 2 and eng.counter("tdf3_pair_image_launches") - p0 == 1, "the pair-image reader did not run"
    eng.set_option("gemm_pair_images", 0)
    p0 = eng.counter("tdf3_pair_image_launches")
    y0, h0 = eng.op_tdf_block(x, w0, sc0, sh0, w1, sc1, sh1, return_hidden=True)
    assert eng.counter("tdf3_pair_image_launches") == p0, "pair image with the option off"
    assert np.isfinite(y1).all() and np.isfinite(h1).all(), "unwritten (NaN canary) elements"
    x64 = x.astype(np.float64)
    h64 = np.maximum(sc0[None, :, None, None] * (x64 @ w0.astype(np.float64).T) + sh0[None, :, None, None], 0)
    ref = np.maximum(sc1[None, :, None, None] * (h64 @ w1.astype(np.float64).T) + sh1[None, :, None, None], 0) + x64
    # the decoded image holds the fp32 activations to 2^-21 of each (row, span)'s largest value (two 11-bit parts, rounding of either)
    # (a span = a column tile of the first linear: 128 or 192 columns, the launcher's choice)
    def worst(cols):
        w = 0.0
        for s0 in range(0, F8, cols):
            a, b = h1[..., s0:s0 + cols].astype(np.float64), h0[..., s0:s0 + cols].astype(np.float64)
            top = np.abs(b).max(axis=-1, keepdims=True)
            w = max(w, float((np.abs(a - b) / np.maximum(top, 1e-300)).max()))
        return w
    wspan = min(worst(128), worst(192))
    assert wspan <= 2.0 ** -21, (worst(128), worst(192))
    # rows: as close to float64 as the block through fp32 activations (same products up to the exponent the parts carry)
    rows = lambda v: np.sqrt(((v - ref) ** 2).sum(-1))
    nrm = np.sqrt(((ref - x64) ** 2).sum(-1)) + 1e-30       # scale of tdf(x), the part the GEMMs produce
    e1, e0 = rows(y1.astype(np.float64)) / nrm, rows(y0.astype(np.float64)) / nrm
    live = nrm > 1e-20
    assert (e1[live] <= 1.5 * e0[live] + 3e-7).all(), (float(e1[live].max()), float(e0[live].max()))
    assert rel_rms(y1, y0) < 2e-7, rel_rms(y1, y0)


@pytest.mark.parametrize("arith", ARITH)
@pytest.mark.parametrize("xs,bar", [(1e-30, None), (1e-35, 1e-4), (1e35, None)])
def test_rowgemm_bf16x6_extreme_magnitudes(A, xs, bar, arith):
    eng = A.Engine(small_cfg(A))
    B, c, T, K, N = 1, 3, 50, 256, 200
    rng = np.random.default_rng(77)
    x = (xs * rng.standard_normal((B, c, T, K))).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    sc, sh = np.ones(c, np.float32), np.zeros(c, np.float32)
    ref = np.maximum(x.astype(np.float64) @ w.astype(np.float64).T, 0)
    try:
        _set_arith(eng, arith)
        n0, h0 = eng.counter("tdf3_launches"), eng.counter("tdf3h_launches")
        y6 = eng.op_tdf(x, w, None, sc, sh, None)
        assert _ran(eng, arith, n0, h0)
        eng.set_option("gemm_bf16x6", 0)
        y32 = eng.op_tdf(x, w, None, sc, sh, None)
    finally:
        eng.set_option("gemm_bf16x6", 1)
    assert np.isfinite(y6).all() and np.isfinite(y32).all()
    # compare in float64 at unit scale (rel_rms clamps its denominator at 1e-30)
    e6, e32 = rel_rms(y6.astype(np.float64) / xs, ref / xs), rel_rms(y32.astype(np.float64) / xs, ref / xs)
    print(f"{arith} row GEMM at |x| ~ {xs:g}: rel-RMS vs float64 {e6:.3e} (fp32-MFMA kernel {e32:.3e})")
    if bar is None:
        assert e6 < 2e-6 and e6 <= 1.25 * e32 + 1e-8, (e6, e32)
    else:
        assert e6 < bar, (e6, e32)


@pytest.mark.parametrize("arith", ARITH)
def test_rowgemm_bf16x6_nonfinite_rows(A, arith):
    """fp16 x 3: the exponents are per row, so no other row can change by a bit either; the poisoned rows are NaN accumulators (Inf 2^e is
    Inf, its low part Inf - Inf) as in bf16 x 6."""
    eng = A.Engine(small_cfg(A))
    B, c, T, K, N = 1, 2, 40, 128, 136
    rng = np.random.default_rng(78)
    clean = rng.standard_normal((B, c, T, K)).astype(np.float32)
    x = clean.copy()
    x[0, 0, 3, 17] = np.inf
    x[0, 0, 9, 0] = -np.inf
    x[0, 1, 5, 127] = np.nan
    x[0, 1, 30, 64] = np.inf
    x[0, 1, 30, 65] = -np.inf
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    sc, sh = np.ones(c, np.float32), np.zeros(c, np.float32)
    try:
        _set_arith(eng, arith)
        y6, y6c = eng.op_tdf(x, w, None, sc, sh, None), eng.op_tdf(clean, w, None, sc, sh, None)
        eng.set_option("gemm_bf16x6", 0)
        y32, y32c = eng.op_tdf(x, w, None, sc, sh, None), eng.op_tdf(clean, w, None, sc, sh, None)
    finally:
        eng.set_option("gemm_bf16x6", 1)
    bad = np.zeros((B, c, T), bool)
    for idx in ((0, 0, 3), (0, 0, 9), (0, 1, 5), (0, 1, 30)):
        bad[idx] = True
    assert np.array_equal(y6[~bad], y6c[~bad]), "a non-finite input leaked into another row"
    assert np.array_equal(y32[~bad], y32c[~bad]), "a non-finite input leaked into another row"
    assert np.isfinite(y6[~bad]).all()
    assert (y6[bad] == 0).all(), "split operands: NaN accumulators through the ReLU maxNum"
    assert not np.isnan(y32[bad]).any() and np.isinf(y32[0, 0, 3]).any() and ((y32[bad] == 0) | np.isinf(y32[bad])).all()


def test_rowgemm_bf16x6_weight_cache_follows_reloads(A):
    """The split image is cached per weight tensor; a second layer uploaded to (possibly) the same address must not see the first's."""
    eng = A.Engine(small_cfg(A))
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 2, 64, 128)).astype(np.float32)
    sc, sh = np.ones(2, np.float32), np.zeros(2, np.float32)
    for seed in range(3):
        w = (np.random.default_rng(seed).standard_normal((128, 128)) / 11.0).astype(np.float32)
        y = eng.op_tdf(x, w, None, sc, sh, None)
        ref = np.maximum(x.astype(np.float64) @ w.astype(np.float64).T, 0)
        assert rel_rms(y, ref) < 1e-6, (seed, rel_rms(y, ref))


# ---------------------------------------------------------------------------
# whole net
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("fname,bias,wseed", [("net_small.npz", False, 3), ("net_small_bias.npz", True, 4)])
def test_net_small_golden(A, golden_dir, fname, bias, wseed):
    g = np.load(os.path.join(golden_dir, fname))
    eng, sd, d = small_engine(A, seed=wseed, bias=bias)
    x = np.random.default_rng(int(g["x_seed"])).standard_normal((2, 4, 32, 16)).astype(np.float32)
    y = eng.net_forward(x)
    assert rel_rms(y, g["y"]) < 2e-5, rel_rms(y, g["y"])


# the ConvTDFNet forms besides BatchNorm / bn > 0 (uvr_lib_v5/mdxnet.py:45-49, modules.py:52-70): goldens written by the reference
# class (tests/golden/make_golden_variants.py)
NET_VARIANTS = {"gn": ("group", 4, False, 11), "gn_bias": ("group", 4, True, 12), "bn0": ("batch", 0, False, 13), "gn_bn0": ("group", 0, True, 14),
                "notdf": ("batch", None, False, 15)}


def variant_engine(A, name, max_batch=0):
    norm, bn, bias, seed = NET_VARIANTS[name]
    d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=bn, bias=bias, norm=norm)
    sd = O.make_convtdf_state(d, seed=seed)
    eng = A.Engine(small_cfg(A, 0.25, False, max_batch))
    eng.load_net(A.NetConfig(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=bn, tdf_bias=bias, norm=norm),
                 A.fold_convtdf_state(sd, d.num_blocks, d.l, tdf_bias=bias))
    return eng, sd, d


@pytest.mark.parametrize("name", list(NET_VARIANTS))
def test_net_variants_golden(A, golden_dir, name):
    g = np.load(os.path.join(golden_dir, "net_variants.npz"))
    eng, sd, d = variant_engine(A, name)
    x = np.random.default_rng(int(g["x_seed"])).standard_normal((2, 4, 32, 16)).astype(np.float32)
    y = eng.net_forward(x)
    assert rel_rms(y, g[name]) < 2e-5, rel_rms(y, g[name])
    assert eng.net_flops(1) == O.net_flops(d)


def test_groupnorm_demix_golden_and_batch_invariance(A, golden_dir):
    """GroupNorm statistics are per chunk: the chunk loop's result must not depend on how many chunks share a net pass."""
    g = np.load(os.path.join(golden_dir, "net_variants.npz"))
    mix = (0.4 * np.random.default_rng(int(g["mix_seed"])).standard_normal((2, int(g["mix_n"])))).astype(np.float32)
    outs = []
    for mb in (1, 3, 64):
        eng, _, _ = variant_engine(A, "gn", max_batch=mb)
        outs.append(eng.demix(mix))
    assert rel_rms(outs[0], g["gn_demix"]) < TOL_STEM, rel_rms(outs[0], g["gn_demix"])
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_groupnorm_net_mid_vs_oracle(A):
    # HQ_3's channel plan with GroupNorm(2, c) on a reduced spectrogram: production tile shapes under the unfused norm passes
    d = O.NetDims(dim_c=4, dim_f=768, dim_t=64, g=48, l=3, num_blocks=11, k=3, bn=8, norm="group")
    sd = O.make_convtdf_state(d, seed=1)
    eng = A.Engine(A.MDXConfig(n_fft=1536, hop_length=256, dim_f=768, segment_size=64))
    eng.load_net(A.NetConfig(dim_c=4, dim_f=768, dim_t=64, g=48, l=3, num_blocks=11, k=3, bn=8, norm="group"), A.fold_convtdf_state(sd, d.num_blocks, d.l))
    x = np.random.default_rng(2).standard_normal((2, 4, 768, 64)).astype(np.float32)
    y = eng.net_forward(x)
    ref = O.convtdf_forward(x, sd, d)
    assert rel_rms(y, ref) < 2e-5, rel_rms(y, ref)


def test_net_mid_vs_oracle(A):
    # HQ_3's channel plan (g=48, 11 blocks, l=3, bn=8) on a reduced spectrogram
    d = O.NetDims(dim_c=4, dim_f=768, dim_t=64, g=48, l=3, num_blocks=11, k=3, bn=8)
    sd = O.make_convtdf_state(d, seed=0)
    eng = A.Engine(A.MDXConfig(n_fft=1536, hop_length=256, dim_f=768, segment_size=64))
    eng.load_net(A.NetConfig(dim_c=4, dim_f=768, dim_t=64, g=48, l=3, num_blocks=11, k=3, bn=8),
                 A.fold_convtdf_state(sd, d.num_blocks, d.l))
    x = np.random.default_rng(1).standard_normal((3, 4, 768, 64)).astype(np.float32)
    y = eng.net_forward(x)
    ref = O.convtdf_forward(x, sd, d)
    assert rel_rms(y, ref) < 2e-5, rel_rms(y, ref)
    assert abs(eng.net_flops(1) / O.net_flops(d) - 1) < 1e-9


# ---------------------------------------------------------------------------
# chunk loop
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,overlap,denoise,match", [
    ("ov25", 0.25, False, False), ("ov25_denoise", 0.25, True, False), ("ov0", 0.0, False, False),
    ("ov75", 0.75, False, False), ("match", 0.25, False, True)])
def test_demix_small_golden(A, golden_dir, name, overlap, denoise, match):
    g = np.load(os.path.join(golden_dir, "demix_small.npz"))
    N = int(g["N"])
    mix = (0.4 * np.random.default_rng(int(g["mix_seed"])).standard_normal((2, N))).astype(np.float32)
    eng, _, _ = small_engine(A, overlap=overlap, denoise=denoise, max_batch=5)
    out = eng.demix(mix, is_match_mix=match)
    assert out.shape == (2, N) and np.isfinite(out).all()
    assert rel_rms(out, g[name]) < TOL_STEM, rel_rms(out, g[name])


@pytest.mark.parametrize("n", [1, 143, 144, 145])
def test_demix_ragged_golden(A, golden_dir, n):
    g = np.load(os.path.join(golden_dir, "demix_small.npz"))
    mix = (0.4 * np.random.default_rng(100 + n).standard_normal((2, n))).astype(np.float32)
    eng, _, _ = small_engine(A)
    out = eng.demix(mix)
    assert out.shape == (2, n)
    assert rel_rms(out, g[f"ragged_n{n}"]) < TOL_STEM


def test_plan_matches_reference_arithmetic(A):
    eng = A.Engine(A.MDXConfig())
    p = eng.plan(10_584_000)
    assert (p["chunk_size"], p["gen_size"], p["pad"], p["padded_len"], p["step"], p["n_chunks"], p["trim"]) == \
        (261120, 254976, 128064, 10_715_136, 195840, 55, 3072)
    assert eng.plan(10_584_000, is_match_mix=True)["n_chunks"] == 42
    for N in (1, 1000, 254976, 254977, 44100 * 30):
        ref = O.chunk_plan(N, O.MDXParams(), False)
        q = eng.plan(N)
        assert (q["chunk_size"], q["gen_size"], q["pad"], q["padded_len"], q["step"], q["n_chunks"]) == \
            (ref[0], ref[1], ref[2], ref[3], ref[4], len(ref[5]))


@pytest.mark.parametrize("overlap", [0.1, 0.9, 0.05, 0.15, 0.2, 0.3, 0.4, 0.6, 0.001, 0.999])
def test_plan_non_dyadic_overlap(A, overlap):
    """step = int((1 - overlap) * chunk_size) in Python doubles (mdx_separator.py:335); float32 overlap would be off by one
    (ADVICE r1): hop 1024 / seg 256 at overlap 0.1 is 235008, not 235007."""
    eng = A.Engine(A.MDXConfig(overlap=overlap))
    for N in (1000, 44100 * 30, 10_584_000):
        ref = O.chunk_plan(N, O.MDXParams(overlap=overlap), False)
        q = eng.plan(N)
        assert (q["step"], q["n_chunks"], q["padded_len"]) == (ref[4], len(ref[5]), ref[3]), (overlap, N, q, ref[4])
    if overlap == 0.1:
        assert eng.plan(1000)["step"] == 235008


@pytest.mark.parametrize("overlap", [0.1, 0.9])
def test_demix_non_dyadic_overlap_vs_oracle(A, overlap):
    eng, sd, d = small_engine(A, overlap=overlap)
    mix = (0.4 * np.random.default_rng(77).standard_normal((2, 3000))).astype(np.float32)
    ref = O.demix(mix, O.MDXParams(n_fft=96, hop_length=16, dim_f=32, segment_size=16, overlap=overlap), O.make_model_run(sd, d))
    out = eng.demix(mix)
    assert rel_rms(out, ref) < TOL_STEM, rel_rms(out, ref)


def test_errors(A):
    eng, _, _ = small_engine(A)
    with pytest.raises(ValueError):
        eng.demix(np.zeros((1, 100), np.float32))
    with pytest.raises(A.AsxError):
        A.Engine(A.MDXConfig(n_fft=6146))             # 3073 does not factor into {2,3,5}
    e2 = A.Engine(small_cfg(A))
    with pytest.raises(A.AsxError):
        e2.demix(np.zeros((2, 100), np.float32))      # weights not committed
    dm = A.MDXDemixer({"model_data": {"compensate": 1.0, "mdx_dim_f_set": 32, "mdx_dim_t_set": 4,
                                       "mdx_n_fft_scale_set": 96}, "torch_device": 0},
                      {"segment_size": 16, "overlap": 0.25, "hop_length": 16, "enable_denoise": False})
    with pytest.raises(ValueError):
        dm.demix(np.zeros((2, 0), np.float32))


def test_stems_small_golden(A, golden_dir):
    g = np.load(os.path.join(golden_dir, "stems_small.npz"))
    mix = (0.8 * np.random.default_rng(int(g["mix_seed"])).standard_normal((2, 2000))).astype(np.float32)
    d = SMALL_DIMS
    dm = A.MDXDemixer({"model_data": {"compensate": float(g["compensate"]), "mdx_dim_f_set": 32, "mdx_dim_t_set": 4,
                                       "mdx_n_fft_scale_set": 96}, "torch_device": 0,
                       "normalization_threshold": 0.9, "amplification_threshold": 0.0},
                      {"segment_size": 16, "overlap": 0.25, "hop_length": 16, "enable_denoise": False, "batch_size": 1},
                      state_dict=O.make_convtdf_state(d, seed=3),
                      net_config=A.NetConfig(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4))
    primary, secondary = dm.separate_stems(mix)
    assert np.array_equal(mix, g["mix_norm"])
    assert rel_rms(primary, g["primary"]) < TOL_STEM
    assert rel_rms(secondary, g["secondary"]) < TOL_STEM


def test_run_model_vs_oracle(A):
    eng, sd, d = small_engine(A)
    p = O.MDXParams(n_fft=96, hop_length=16, dim_f=32, segment_size=16)
    w = (0.3 * np.random.default_rng(9).standard_normal((3, 2, 240))).astype(np.float32)
    got = eng.run_model(w)
    ref = O.run_model(w, p, O.make_model_run(sd, d))
    assert rel_rms(got, ref) < TOL_STEM


def test_fft3_fast_path_vs_generic_and_oracle(A, monkeypatch):
    """n_fft 6144 / hop 1024 runs the three-pass register FFT with the frame overlap-add fused into the inverse
    (csrc/kernels_fft3.h); ASX_FFT3=0 keeps the generic six-pass kernels + frame buffer + ola_kernel.  Both against the oracle
    (STFT -> bins [0, 3) zeroed -> iSTFT, the match-mix pass of run_model) on a 3-chunk batch, at a short segment too (T = 40:
    one frame group, no seams) and at T = 70 (two groups, the second longer than G)."""
    for seg, dim_f in ((256, 3072), (40, 3072), (70, 3072), (70, 2048)):
        C = 1024 * (seg - 1)
        w = (0.3 * np.random.default_rng(seg).standard_normal((3, 2, C))).astype(np.float32)
        ref = O.run_model(w, O.MDXParams(segment_size=seg, dim_f=dim_f), None, is_match_mix=True)
        monkeypatch.setenv("ASX_FFT3", "1")
        monkeypatch.setenv("ASX_FFT3P", "1")
        fast = A.Engine(A.MDXConfig(segment_size=seg, dim_f=dim_f)).run_model(w, is_match_mix=True)
        monkeypatch.setenv("ASX_FFT3P", "0")      # the inverse without the LDS-DMA spectrum prefetch (the denoise path's kernel)
        fast0 = A.Engine(A.MDXConfig(segment_size=seg, dim_f=dim_f)).run_model(w, is_match_mix=True)
        monkeypatch.setenv("ASX_FFT3", "0")
        slow = A.Engine(A.MDXConfig(segment_size=seg, dim_f=dim_f)).run_model(w, is_match_mix=True)
        assert rel_rms(fast, ref) < 5e-6, (seg, rel_rms(fast, ref))
        assert rel_rms(fast0, ref) < 5e-6, (seg, rel_rms(fast0, ref))
        assert rel_rms(slow, ref) < 5e-6, (seg, rel_rms(slow, ref))
        assert rel_rms(fast, slow) < 5e-6 and max_abs(fast, slow) < 2e-5, (seg, max_abs(fast, slow))
        assert rel_rms(fast, fast0) < 1e-6, (seg, rel_rms(fast, fast0))
    monkeypatch.delenv("ASX_FFT3P")
    monkeypatch.delenv("ASX_FFT3")


def test_fft3_song_mode_edges_vs_oracle(A, monkeypatch):
    """The multi-frame forward kernel maps every hop through the song position (zero outside the song, reflection at the chunk
    ends, unaligned starts); the inverse groups frames by T alone.  Match-mix demix (STFT -> zero bins -> iSTFT -> Hann fold, no
    net) of mixes shorter than a chunk, of odd length, and at a short segment, against the oracle and the generic kernels."""
    for seg, N in ((256, 100_000), (256, 300_001), (256, 523_457), (8, 30_011), (20, 44_100)):
        mix = O.synth_mix(N, seed=N % 97)
        ref = O.demix(mix, O.MDXParams(segment_size=seg), None, is_match_mix=True)
        monkeypatch.setenv("ASX_FFT3", "1")
        fast = A.Engine(A.MDXConfig(segment_size=seg)).demix(mix, is_match_mix=True)
        monkeypatch.setenv("ASX_FFT3", "0")
        slow = A.Engine(A.MDXConfig(segment_size=seg)).demix(mix, is_match_mix=True)
        assert fast.shape == ref.shape
        assert rel_rms(fast, ref) < 1e-5, (seg, N, rel_rms(fast, ref))
        assert rel_rms(fast, slow) < 5e-6 and max_abs(fast, slow) < 2e-5, (seg, N, max_abs(fast, slow))
    monkeypatch.delenv("ASX_FFT3")


def test_batching_is_invisible(A):
    # results must not depend on how many chunks share a device batch (reference: batch_size has no effect)
    mix = (0.4 * np.random.default_rng(77).standard_normal((2, 5000))).astype(np.float32)
    outs = []
    for mb in (1, 3, 64):
        eng, _, _ = small_engine(A, max_batch=mb)
        outs.append(eng.demix(mix))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_sharded_chunks_equal_single_pass(A):
    # asx_demix_chunks_dev over two chunk ranges + asx_finalize_dev == asx_demix_dev, bit for bit
    import torch
    eng, _, _ = small_engine(A)
    N = 5000
    mix = torch.tensor((0.4 * np.random.default_rng(78).standard_normal((2, N))).astype(np.float32)).cuda()
    out1 = torch.empty_like(mix)
    s = torch.cuda.current_stream().cuda_stream
    eng.demix_dev(mix.data_ptr(), N, out1.data_ptr(), stream=s)
    p = eng.plan(N)
    nk, C = p["n_chunks"], p["chunk_size"]
    co = torch.zeros((nk, 2, C), dtype=torch.float32, device="cuda")
    half = nk // 2
    eng.demix_chunks_dev(mix.data_ptr(), N, 0, half, co.data_ptr(), stream=s)
    eng.demix_chunks_dev(mix.data_ptr(), N, half, nk, co[half:].data_ptr(), stream=s)
    out2 = torch.empty_like(mix)
    eng.finalize_dev(co.data_ptr(), N, out2.data_ptr(), stream=s)
    torch.cuda.synchronize()
    assert torch.equal(out1, out2)


def test_demix_dev_is_capturable_into_a_hip_graph(A):
    """include/asx.h: once the workspace has its size, asx_demix_dev only enqueues kernels on the caller's stream -- no host copy,
    no synchronisation -- so a whole demix can be captured into a hipGraph and replayed.  Generic FFT path (small net) and the
    6144 / 1024 fast path (match-mix pass, no net); the replayed result equals the directly launched one bit for bit."""
    import torch
    cases = []
    eng, _, _ = small_engine(A)
    cases.append((eng, 5000, 0))
    cases.append((A.Engine(A.MDXConfig(segment_size=40)), 90_000, 1))       # n_fft 6144 / hop 1024, ASX_FLAG_MATCH_MIX
    for eng, N, flags in cases:
        mix = torch.tensor((0.4 * np.random.default_rng(N).standard_normal((2, N))).astype(np.float32)).cuda()
        direct = torch.empty_like(mix)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            eng.demix_dev(mix.data_ptr(), N, direct.data_ptr(), is_match_mix=bool(flags), stream=side.cuda_stream)   # sizes the workspace
        torch.cuda.synchronize()
        replayed = torch.zeros_like(mix)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            eng.demix_dev(mix.data_ptr(), N, replayed.data_ptr(), is_match_mix=bool(flags),
                          stream=torch.cuda.current_stream().cuda_stream)
        assert float(replayed.abs().sum()) == 0.0          # nothing ran during capture
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(direct, replayed)
        mix.mul_(0.5)                                      # same buffers, new content: the graph is reusable
        graph.replay()
        torch.cuda.synchronize()
        again = torch.empty_like(mix)
        eng.demix_dev(mix.data_ptr(), N, again.data_ptr(), is_match_mix=bool(flags), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(again, replayed)


def test_captured_graph_survives_other_engines_loading_and_dying(A):
    """the split weight images of the bf16 x 6 kernels were ONE process-wide cache that every engine's load or
    destroy flushed -- a hipGraph captured on engine A embeds those image pointers, so engine B's load freed them under it.  They are
    per-engine now: capture a demix of A (TDF linears wide enough for tdf3_kernel), then load, run and destroy two other engines
    (whose own images would reuse the freed memory), then replay."""
    import torch
    d = O.NetDims(dim_c=4, dim_f=768, dim_t=64, g=8, l=1, num_blocks=3, k=3, bn=8)
    cfg = A.MDXConfig(n_fft=1536, hop_length=256, dim_f=768, segment_size=64)
    ncfg = A.NetConfig(dim_c=4, dim_f=768, dim_t=64, g=8, l=1, num_blocks=3, k=3, bn=8)

    def engine(seed):
        eng = A.Engine(cfg)
        eng.load_net(ncfg, A.fold_convtdf_state(O.make_convtdf_state(d, seed=seed), d.num_blocks, d.l))
        return eng
    a = engine(1)
    N = 40_000
    mix = torch.tensor((0.4 * np.random.default_rng(N).standard_normal((2, N))).astype(np.float32)).cuda()
    direct, replayed = torch.empty_like(mix), torch.zeros_like(mix)
    side = torch.cuda.Stream()
    n0 = a.counter("tdf3_launches")
    with torch.cuda.stream(side):
        a.demix_dev(mix.data_ptr(), N, direct.data_ptr(), stream=side.cuda_stream)       # sizes the workspace, builds A's split images
    torch.cuda.synchronize()
    assert a.counter("tdf3_launches") > n0, "the TDF linears of this net did not run on tdf3_kernel: the test would prove nothing"
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        a.demix_dev(mix.data_ptr(), N, replayed.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    other = torch.empty_like(mix)
    for seed in (2, 3):                                   # other engines come, build their own images, and go
        b = engine(seed)
        b.demix_dev(mix.data_ptr(), N, other.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert not torch.equal(other, direct)
        b.close()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(direct, replayed)
    a.close()


# ---------------------------------------------------------------------------
# HQ_3 geometry (the metric configuration), bounded so the CPU oracle finishes
# ---------------------------------------------------------------------------
def test_demix_hq3_excerpt_vs_oracle(A):
    # 12 s of 44.1 kHz stereo = 3 chunks through the full-size HQ_3-shaped net
    d = O.NetDims()
    sd = O.make_convtdf_state(d, seed=0)
    N = 44100 * 12
    mix = O.synth_mix(N, seed=0)
    eng = A.Engine(A.MDXConfig(max_batch=2))
    eng.load_net(A.NetConfig(), A.fold_convtdf_state(sd, d.num_blocks, d.l))
    got = eng.demix(mix)
    ref = O.demix(mix, O.MDXParams(), O.make_model_run(sd, d))
    e = rel_rms(got, ref)
    print("HQ_3 excerpt rel-RMS:", e, "ref rms", float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))))
    assert e < TOL_STEM, e


def test_match_mix_full_song_vs_oracle(A):
    # full 4-minute geometry, no net: 42 chunks of STFT -> zero bins -> iSTFT -> Hann fold
    N = 10_584_000
    mix = O.synth_mix(N, seed=0)
    eng = A.Engine(A.MDXConfig())
    got = eng.demix(mix, is_match_mix=True)
    ref = O.demix(mix, O.MDXParams(), None, is_match_mix=True)
    assert rel_rms(got, ref) < 1e-5, rel_rms(got, ref)


def test_full_song_properties(A):
    # size-independent properties at BASELINE's full size (55 chunks, full net):
    #  * finite output of the right shape
    #  * the net is positively homogeneous only through ReLU, but the chunk loop is linear in the
    #    net output: demix with denoise on a net whose output is odd in its input equals the plain demix.
    #    Cheaper and exact: output does not depend on max_batch (chunks are independent).
    d = O.NetDims()
    sd = O.make_convtdf_state(d, seed=0)
    N = 44100 * 60
    mix = O.synth_mix(N, seed=3)
    outs = []
    for mb in (16, 5):
        eng = A.Engine(A.MDXConfig(max_batch=mb))
        eng.load_net(A.NetConfig(), A.fold_convtdf_state(sd, d.num_blocks, d.l))
        outs.append(eng.demix(mix))
        eng.close()
    assert outs[0].shape == (2, N) and np.isfinite(outs[0]).all()
    assert np.array_equal(outs[0], outs[1])


# ---------------------------------------------------------------------------
# model-file loading (ONNX) and the device-side stem algebra
# ---------------------------------------------------------------------------
def test_onnx_model_file_end_to_end(A, golden_dir):
    # an .onnx written by torch's exporter from the reference ConvTDFNet -> reader -> engine -> golden demix
    g = np.load(os.path.join(golden_dir, "demix_small.npz"))
    N = int(g["N"])
    mix = (0.4 * np.random.default_rng(int(g["mix_seed"])).standard_normal((2, N))).astype(np.float32)
    dm = A.MDXDemixer({"model_data": {"compensate": 1.0, "mdx_dim_f_set": 32, "mdx_dim_t_set": 4,
                                       "mdx_n_fft_scale_set": 96}, "torch_device": 0,
                       "model_path": os.path.join(golden_dir, "net_small.onnx")},
                      {"segment_size": 16, "overlap": 0.25, "hop_length": 16, "enable_denoise": False})
    assert (dm.net_config.g, dm.net_config.l, dm.net_config.num_blocks, dm.net_config.bn) == (8, 2, 5, 4)
    out = dm.demix(mix)
    assert rel_rms(out, g["ov25"]) < TOL_STEM, rel_rms(out, g["ov25"])


def test_separate_on_device_bit_exact_algebra(A):
    # normalisation and the stem algebra are single-rounding float32 ops: identical to numpy given the same demix
    eng, sd, d = small_engine(A)
    rng = np.random.default_rng(5)
    for scale in (0.8, 0.2):                      # with and without normalisation kicking in
        mix = (scale * rng.standard_normal((2, 2500))).astype(np.float32)
        ref_mix = mix.copy()
        peak = np.abs(ref_mix).max()
        ref_mix = O.normalize(ref_mix, 0.9, 0.0)
        dem = eng.demix(ref_mix)
        ref_primary = (dem * peak).T
        ref_secondary = (-ref_primary * 1.035) + ref_mix.T
        primary, secondary = eng.separate(mix, 0.9, 0.0, 1.035)
        assert np.array_equal(mix, ref_mix)
        assert np.array_equal(primary, ref_primary)
        assert np.array_equal(secondary, ref_secondary)


def _set_or_skip(eng, key, value):
    """Options that name a superseded kernel generation exist in experimental builds only (python build.py --experimental): skip elsewhere."""
    import audio_separator_amd as A_
    try:
        eng.set_option(key, value)
    except A_.AsxError as e:
        if "experimental" in str(e):
            pytest.skip(f"{key} = {value}: {e}")
        raise


# ---------------------------------------------------------------------------
# Winograd F(2x2, 3x3) option for the 3x3 convolutions
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("B,cin,cout,T,F", [(1, 48, 48, 16, 128), (2, 96, 96, 8, 64), (1, 32, 32, 24, 200),
                                            (2, 8, 8, 16, 32), (1, 24, 24, 4, 8), (1, 144, 144, 8, 96),
                                            (1, 80, 80, 10, 36), (1, 288, 288, 8, 96), (1, 50, 70, 6, 44),
                                            (1, 16, 16, 5, 8), (2, 8, 12, 7, 12), (1, 4, 48, 9, 100), (1, 48, 4, 3, 4)])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_conv3x3_winograd(A, B, cin, cout, T, F, mode):
    # mode 3 is the engine's default; 0 = the direct MFMA kernel (kept covered here now that it is not the default), 1 / 2 = the
    # earlier Winograd generations
    eng = A.Engine(small_cfg(A))
    _set_or_skip(eng, "winograd", mode)
    rng = np.random.default_rng(cin * 1000 + cout + T + 7)
    x = rng.standard_normal((B, cin, T, F)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    y = eng.op_conv("conv3x3", x, w, b)
    ref = _torch_ref("conv3x3", x, w, b)
    assert np.isfinite(y).all(), "unwritten (NaN canary) output elements"
    assert max_abs(y, ref) < 5e-5, (max_abs(y, ref), rel_rms(y, ref))


@pytest.mark.parametrize("B,cin,cout,T,F", [(1, 48, 48, 16, 128), (2, 96, 96, 8, 64), (1, 48, 96, 64, 64), (3, 96, 48, 33, 32),
                                            (1, 44, 50, 7, 96), (1, 90, 4, 2, 32), (1, 96, 144, 1, 64), (2, 48, 48, 130, 3072 // 8),
                                            (1, 144, 144, 9, 96), (2, 144, 50, 16, 32), (1, 48, 20, 5, 64)])
def test_conv3x3_winograd_stationary(A, B, cin, cout, T, F, variant=1, relu=True):
    """conv_winos_kernel (kernels_winos.h): weights resident in registers, positions split over eight waves, column strips of 32
    pixels walked one tile row per step -- selectable for 3x3 layers with 41..48 / 89..96 input channels on planes whose width is
    a multiple of 32 (option "winograd_stationary").  Against torch, and against conv_wino3_kernel (option off)."""
    eng = A.Engine(small_cfg(A))
    assert eng.option("winograd") == 3
    assert eng.option("winograd_stationary") == 0       # measured slower than conv_wino3_kernel: selectable, not the default
    _set_or_skip(eng, "winograd_stationary", variant)
    rng = np.random.default_rng(cin * 1000 + cout + T + 11)
    x = rng.standard_normal((B, cin, T, F)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    y = eng.op_conv("conv3x3", x, w, b, relu=relu)
    ref = _torch_ref("conv3x3", x, w, b, relu=relu)
    assert np.isfinite(y).all(), "unwritten (NaN canary) output elements"
    assert max_abs(y, ref) < 5e-5, (max_abs(y, ref), rel_rms(y, ref))
    eng.set_option("winograd_stationary", 0)
    y3 = eng.op_conv("conv3x3", x, w, b, relu=relu)
    assert max_abs(y, y3) < 2e-5, max_abs(y, y3)


@pytest.mark.parametrize("B,cin,cout,T,F,relu", [(1, 144, 144, 8, 96, True), (2, 96, 96, 10, 44, True), (1, 288, 288, 8, 96, True), (1, 64, 20, 16, 64, False),
                                                 (1, 80, 80, 9, 36, True), (3, 192, 192, 5, 33 * 4, True), (1, 240, 240, 16, 192, True), (2, 160, 50, 3, 8, False)])
@pytest.mark.parametrize("arith", ARITH)
def test_conv3x3_winograd_bf16x6(A, B, cin, cout, T, F, relu, arith):
    """conv_wino6_kernel (csrc/kernels_wino6.h, round 5: the default for 3x3 TFC layers with >= 144 input channels): Winograd
    F(2x2,3x3) with the sixteen transform-domain GEMMs on split operands -- six bf16 MFMA products on exact three-way splits, or
    (gemm_f16x3, the default) three fp16 products on block-scaled two-way splits.  Against torch, against conv_wino3_kernel on the same
    layer, with proof of which kernel and which arithmetic ran; ragged planes, channel counts off the 32 / 48 grids, border tiles,
    several batch items."""
    eng = A.Engine(small_cfg(A))
    assert eng.option("winograd") == 3 and eng.option("winograd_bf16x6") == 144
    eng.set_option("winograd_bf16x6", 64)
    eng.set_option("conv_direct_f16x3", 0)             # (round 6: the direct kernel would take the 144-channel case first)
    eng.set_option("gemm_f16x3", 1 if arith == "f16x3" else 0)
    h0 = eng.counter("wino6h_launches")
    rng = np.random.default_rng(cin * 1000 + cout + T + 13)
    x = rng.standard_normal((B, cin, T, F)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    n0 = eng.counter("wino6_launches")
    y = eng.op_conv("conv3x3", x, w, b, relu=relu)
    assert eng.counter("wino6_launches") == n0 + 1, "conv_wino6_kernel did not run"
    assert eng.counter("wino6h_launches") - h0 == (1 if arith == "f16x3" else 0), "the other arithmetic of conv_wino6_kernel ran"
    assert np.array_equal(y, eng.op_conv("conv3x3", x, w, b, relu=relu)), "not deterministic"
    n0 += 1
    ref = _torch_ref("conv3x3", x, w, b, relu=relu)
    assert np.isfinite(y).all(), "unwritten (NaN canary) output elements"
    assert max_abs(y, ref) < 5e-5, (max_abs(y, ref), rel_rms(y, ref))
    eng.set_option("winograd_bf16x6", 0)
    y3 = eng.op_conv("conv3x3", x, w, b, relu=relu)
    assert eng.counter("wino6_launches") == n0 + 1, "the fp32 run went through conv_wino6_kernel"
    assert max_abs(y, y3) < 2e-5, max_abs(y, y3)
    # float64 reference: at least as close as the fp32 Winograd kernel (a reduced-precision shortcut would fail this by 100x)
    import torch
    r64 = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1)
    r64 = (torch.relu(r64) if relu else r64).numpy()
    e6, e3 = rel_rms(y, r64), rel_rms(y3, r64)
    assert e6 < 1e-6 and e6 <= 1.25 * e3 + 1e-8, (e6, e3)


@pytest.mark.parametrize("bits,order", [(12, "decay"), (30, "decay"), (60, "decay"), (30, "grow"), (60, "grow")])
def test_conv3x3_winograd_f16x3_block_exponent(A, bits, order):
    """The running block exponent of conv_wino6_kernel<H> under stress: one power-of-two exponent per tile row and 32-channel chunk of
    the transformed input, carried along the input channels.  The input channels decay (or grow) by 2^bits across the layer and half of the
    output channels look at the QUIET half of the input channels only.  The exponent follows the magnitude both ways (round 6: it used to drop
    only, so a decaying layer carried its loudest chunk's exponent to the end and the quiet half lost bits / 2 - 12 bits): both halves of
    the output fp32-grade at any spread."""
    import torch
    B, cin, cout, T, F = 1, 192, 64, 16, 64
    eng = A.Engine(small_cfg(A))
    eng.set_option("winograd_bf16x6", 64)
    eng.set_option("conv_direct_f16x3", 0)
    rng = np.random.default_rng(bits * 3 + len(order))
    e = bits * np.arange(cin) / (cin - 1)
    mag = np.exp2(-e if order == "decay" else e - bits).astype(np.float32)                       # loudest channel at 1.0 either way
    x = rng.standard_normal((B, cin, T, F)).astype(np.float32) * mag[None, :, None, None]
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    quiet = np.arange(cin) >= cin // 2 if order == "decay" else np.arange(cin) < cin // 2
    w[: cout // 2, ~quiet] = 0                                                                   # output channels 0 .. 31: quiet inputs only
    b = np.zeros(cout, np.float32)
    n0 = eng.counter("wino6h_launches")
    y = eng.op_conv("conv3x3", x, w, b, relu=False)
    assert eng.counter("wino6h_launches") == n0 + 1, "conv_wino6_kernel<H> did not run"
    assert np.isfinite(y).all()
    r64 = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), padding=1).numpy()
    eq, el = rel_rms(y[:, : cout // 2], r64[:, : cout // 2]), rel_rms(y[:, cout // 2:], r64[:, cout // 2:])
    print(f"wino6<H>, channels {order} by 2^{bits}: rel-RMS vs float64 {eq:.2e} (outputs of the quiet half) / {el:.2e} (the others)")
    assert eq < 5e-7 and el < 5e-7, (eq, el)


@pytest.mark.parametrize("B,T,F,relu,spread,c", [(1, 16, 128, True, 0.0, 48), (2, 8, 64, True, 0.0, 48), (3, 10, 96, False, 0.0, 48), (1, 4, 32, True, 0.0, 48),
                                                 (1, 37, 96, False, 3.0, 48), (2, 67, 64, False, 8.0, 48), (2, 5, 1088, True, 2.0, 48), (1, 130, 3072 // 8, True, 1.0, 48),
                                                 (2, 8, 64, True, 0.0, 96), (1, 18, 96, False, 2.0, 96), (1, 128, 512, True, 1.0, 96), (2, 64, 256, True, 1.0, 96),
                                                 (1, 16, 256, True, 1.0, 144)])
def test_conv3x3_direct_f16x3(A, B, T, F, relu, spread, c):
    """conv3h_kernel (csrc/kernels_conv3h.h, round 6: the default for the 48 -> 48 layers, level 0 of the HQ_3 geometry): the 3x3 convolution
    as a direct implicit GEMM on the fp16 pipe -- two-part operands, three products, weights resident in LDS, input rows in a ring walked down T,
    one running power-of-two exponent per walk with accumulator rescales where it moves.  Against torch and float64, against conv_wino3_kernel
    on the same layer, with proof of which kernel ran; ragged T, one-tile planes, several batch items, magnitudes that vary by `spread` decades
    over the plane (the exponent moves: rescale paths), a folded-BatchNorm-like spread of the output channels' weight scales.  Layers of 96 / 144
    channels run as 4 / 9 launches over 48-channel slices (the input slices summed through the output: accumulate mode), on the narrower walks
    (bands of 16 / 8 strips, several segments of T)."""
    import torch
    eng = A.Engine(small_cfg(A))
    assert eng.option("winograd") == 3 and eng.option("conv_direct_f16x3") == 144 and eng.option("gemm_f16x3") == 1
    rng = np.random.default_rng(T * 1000 + F + 17 + c)
    x = rng.standard_normal((B, c, T, F)).astype(np.float32)
    if spread:
        tt, ff = np.meshgrid(np.arange(T), np.arange(F), indexing="ij")
        x *= (10.0 ** (spread * np.sin(0.013 * ff) * np.cos(0.21 * tt))).astype(np.float32)[None, None]
    b = rng.standard_normal(c).astype(np.float32)
    w = (rng.standard_normal((c, c, 3, 3)) / np.sqrt(9 * c)).astype(np.float32)
    w *= (10.0 ** rng.uniform(-1, 1, size=(c, 1, 1, 1))).astype(np.float32)
    n0 = eng.counter("conv3h_launches")
    y = eng.op_conv("conv3x3", x, w, b, relu=relu)
    assert eng.counter("conv3h_launches") == n0 + (c // 48) ** 2, "conv3h_kernel did not run"
    assert np.array_equal(y, eng.op_conv("conv3x3", x, w, b, relu=relu)), "not deterministic"
    assert np.isfinite(y).all(), "unwritten (NaN canary) output elements"
    eng.set_option("conv_direct_f16x3", 0)
    n0 = eng.counter("conv3h_launches")
    y3 = eng.op_conv("conv3x3", x, w, b, relu=relu)
    assert eng.counter("conv3h_launches") == n0, "the fp32 run went through conv3h_kernel"
    r64 = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1)
    # the bar of an fp32 chain: relative to sum |w x| over the taps (what the largest product of a sum is rounded against)
    mag = torch.nn.functional.conv2d(torch.from_numpy(x).double().abs(), torch.from_numpy(w).double().abs(), torch.from_numpy(b).double().abs(), padding=1).numpy()
    r64 = (torch.relu(r64) if relu else r64).numpy()
    assert (np.abs(y - r64) <= 2e-6 * mag + 1e-30).all(), float((np.abs(y - r64) / (mag + 1e-30)).max())
    e6, e3 = rel_rms(y, r64), rel_rms(y3, r64)
    assert e6 < 1e-6 and e6 <= 1.25 * e3 + 1e-8, (e6, e3)
    eng.set_option("conv_direct_f16x3", 48)            # the threshold is a channel count: a wider layer stays off the kernel
    n0 = eng.counter("conv3h_launches")
    eng.op_conv("conv3x3", x, w, b, relu=relu)
    assert eng.counter("conv3h_launches") - n0 == (1 if c == 48 else 0)


@pytest.mark.parametrize("case", ["inf", "nan", "huge", "tiny", "zero"])
def test_conv3x3_direct_f16x3_edge_values(A, case):
    """Non-finite and extreme inputs of conv3h_kernel: an Inf / NaN pixel poisons outputs (as it does in the reference's fp32 convolution --
    here possibly its whole 4 x 32 tile and the walk's following rows up to the next exponent move), never hangs or corrupts other batch items;
    planes of 1e30 or 1e-30 keep fp32-grade relative accuracy (the exponent follows them); an all-zero plane gives the bias."""
    import torch
    eng = A.Engine(small_cfg(A))
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 48, 12, 64)).astype(np.float32)
    b = rng.standard_normal(48).astype(np.float32)
    w = (rng.standard_normal((48, 48, 3, 3)) / np.sqrt(9 * 48)).astype(np.float32)
    if case == "inf":
        x[0, 3, 5, 40] = np.inf
    elif case == "nan":
        x[0, 7, 2, 3] = np.nan
    elif case == "huge":
        x *= np.float32(1e30)
        b[:] = 0
    elif case == "tiny":
        x *= np.float32(1e-30)
        b[:] = 0
    else:
        x[:] = 0
    n0 = eng.counter("conv3h_launches")
    y = eng.op_conv("conv3x3", x, w, b, relu=False)
    assert eng.counter("conv3h_launches") == n0 + 1
    r64 = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1).numpy()
    if case in ("inf", "nan"):
        assert np.isfinite(y[1]).all() and rel_rms(y[1], r64[1]) < 1e-6, "the clean batch item was touched"
        bad = ~np.isfinite(r64[0])
        assert (~np.isfinite(y[0][bad])).all(), "a non-finite reference output came back finite"
        assert np.isfinite(y[0][~bad]).all() and rel_rms(y[0][~bad], r64[0][~bad]) < 1e-6, "outputs the Inf / NaN cannot reach were touched"
    elif case == "zero":
        assert np.array_equal(y, np.broadcast_to(b[None, :, None, None], y.shape))
    else:
        assert np.isfinite(y).all() and rel_rms(y, r64) < 1e-6, rel_rms(y, r64)


_HQ3_EXCERPT = {}


def _hq3_excerpt():
    if not _HQ3_EXCERPT:
        d = O.NetDims()
        sd = O.make_convtdf_state(d, seed=0)
        mix = O.synth_mix(44100 * 12, seed=0)
        _HQ3_EXCERPT.update(d=d, sd=sd, mix=mix, ref=O.demix(mix, O.MDXParams(), O.make_model_run(sd, d)))
    return _HQ3_EXCERPT


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 30, 36, 306, 366, 3000, 3048])
def test_winograd_hq3_excerpt_vs_oracle(A, mode):
    # 3 = the default (round 6: conv3h_kernel -- direct fp16 x 3 -- on the levels up to 144 channels, 1 / 4 / 9 launches per layer over 48-channel
    # slices; conv_wino6_kernel from there up); 3000 = conv3h_kernel off (conv_wino3_kernel on levels 0 / 1, conv_wino6_kernel from 144 channels:
    # round 5's default); 3048 = conv3h_kernel on level 0 only; 30 = the weight-stationary kernel on the two outer levels; 36 = the wino6 threshold
    # at 64 channels (conv3h_kernel still takes its levels first); 306 = wino6 off; 366 = every split-operand kernel on the bf16 x 6 arithmetic
    # (gemm_f16x3 = 0: no conv3h_kernel; conv_wino6_kernel from 64 channels); 0 / 1 / 2 = the direct fp32 kernel / the earlier Winograd generations
    c = _hq3_excerpt()
    d, sd, mix, ref = c["d"], c["sd"], c["mix"], c["ref"]
    eng = A.Engine(A.MDXConfig(max_batch=2))
    # launches per net pass (6 layers per level, 3 at the bottleneck level 5): conv3h_kernel 6 x 1 (level 0) + 6 x 4 (level 1) + 6 x 9 (level 2)
    want3h = {3: 84, 36: 84, 306: 84, 30: 54, 3000: 0, 3048: 6, 366: 0, 0: 0, 1: 0, 2: 0}[mode]
    want6 = {3: 15, 36: 15, 306: 0, 366: 27, 3000: 21, 3048: 21}.get(mode)   # conv_wino6_kernel: levels 3 .. 5 = 15, 2 .. 5 = 21, 1 .. 5 = 27
    if mode == 3000:
        mode = 3
        eng.set_option("conv_direct_f16x3", 0)
    elif mode == 3048:
        mode = 3
        eng.set_option("conv_direct_f16x3", 48)
    h3 = mode != 366
    if mode == 366:
        mode = 36
        eng.set_option("gemm_f16x3", 0)
    if mode == 30:
        mode = 3
        _set_or_skip(eng, "winograd_stationary", 1)
    elif mode == 36:
        mode = 3
        eng.set_option("winograd_bf16x6", 64)
    elif mode == 306:
        mode = 3
        eng.set_option("winograd_bf16x6", 0)
    _set_or_skip(eng, "winograd", mode)
    assert eng.option("winograd") == mode
    eng.load_net(A.NetConfig(), A.fold_convtdf_state(sd, d.num_blocks, d.l))
    n6, n6h, n3h = eng.counter("wino6_launches"), eng.counter("wino6h_launches"), eng.counter("conv3h_launches")
    got = eng.demix(mix)
    passes = -(-eng.plan(mix.shape[1])["n_chunks"] // 2)            # max_batch = 2
    assert eng.counter("conv3h_launches") - n3h == want3h * passes, (eng.counter("conv3h_launches") - n3h, want3h, passes)
    if want6 is not None:
        assert eng.counter("wino6_launches") - n6 == want6 * passes, (eng.counter("wino6_launches") - n6, want6, passes)
        assert eng.counter("wino6h_launches") - n6h == (want6 * passes if h3 else 0), (eng.counter("wino6h_launches") - n6h, want6, passes, h3)
    e = rel_rms(got, ref)
    print("HQ_3 excerpt rel-RMS (winograd):", e)
    assert e < TOL_STEM, e


@pytest.mark.gpu
@pytest.mark.parametrize("peak,max_peak,min_peak", [(1.7, 0.9, None), (0.3, 0.9, 0.5), (0.5, 0.9, 0.0), (0.95, 1.0, None)])
def test_pcm16_writer_edge_bit_exact(peak, max_peak, min_peak):
    """spec_utils.normalize + (stem * 32767).astype(np.int16) + interleave (common_separator.py:309-337), bit for bit"""
    import audio_separator_amd as A
    eng = A.Engine(A.MDXConfig(n_fft=64, hop_length=16, dim_f=32, segment_size=8))
    rng = np.random.default_rng(int(peak * 100))
    stem = rng.standard_normal((50001, 2)).astype(np.float32)
    stem *= np.float32(peak) / np.abs(stem).max()
    w = stem.copy()
    maxv = np.abs(w).max()                       # spec_utils.normalize (spec_utils.py:99-115)
    if maxv > max_peak:
        w *= max_peak / maxv
    elif min_peak is not None and maxv < min_peak:
        w *= min_peak / maxv
    want = (w * 32767).astype(np.int16)
    got, pk = eng.pcm16(stem, max_peak, min_peak)
    assert got.dtype == np.int16 and got.shape == (50001, 2)
    assert np.array_equal(got, want)
    assert abs(pk - np.abs(w).max()) <= 1e-6 * max(1.0, pk)
