"""CPU checks of conv3h_kernel (csrc/kernels_conv3h.h, the direct fp16 x 3 convolution of the 48-channel level):
  * its LDS layout claims against the lane groups `ds_read_b128` / `ds_write_b128` are serviced in on gfx950 (MI355X_MICROARCH.md, LDS table);
  * the producer's item -> (row, channel group, column quad) map covers the 4 x 40 x 48 block exactly once and pairs lanes on 32 contiguous bytes;
  * the host weight packer (cut out of the header, compiled with g++): every weight lands in the fragment slot the kernel reads it from, h + l
    reproduce the scaled weight to 2^-22, one exponent per OUTPUT channel;
  * a numpy emulation of the kernel's arithmetic -- blocks of four rows, running exponent with its three rules, two-part split, three products per
    multiply-add in kernel-row order with the accumulator rescale at kernel-row boundaries -- against a float64 convolution on planes whose magnitude
    moves by decades down T (the exponent drops, rises and hits its bounded-rise rule).  What the block exponent does NOT cover (and nothing here
    asks of it): rows of ONE four-row block more than 2^25 apart in magnitude -- the quiet row is then kept to an absolute 2^-25 of the block's range."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "python-audio-separator_amd", "csrc", "kernels_conv3h.h")

C, CG8, IW, PSTR = 48, 6, 34, 96
ROWB = IW * PSTR
KGY, SPK = 18, 5
WKY = 4 * 6144 + 3 * 2 * 512

B128_READ_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
                    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
                    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
                    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
B128_WRITE_GROUPS = [list(range(8 * g, 8 * g + 8)) for g in range(8)]


def worst_conflict(addr_of_lane, groups, nbanks):
    """largest number of DISTINCT 16-byte accesses that meet on one bank inside a service group (1 = conflict-free; identical addresses broadcast)"""
    worst = 0
    for grp in groups:
        banks = {}
        for lane in grp:
            a = addr_of_lane(lane)
            if a is None:
                continue
            for w in range(4):
                banks.setdefault(((a + 4 * w) // 4) % nbanks, set()).add(a)
        worst = max(worst, max((len(v) for v in banks.values()), default=0))
    return worst


def test_x_fragment_reads_are_conflict_free():
    # stage (ky, sg), pixel tile qq, part: lane (li, g) reads 16 bytes at row * ROWB + (kx + li + 16 qq) * 96 + cig * 16, k group kk = 4 sg + g = (kx, cig)
    for sg in range(SPK):
        for qq in range(2):
            def rd(lane, sg=sg, qq=qq):
                li, g = lane & 15, lane >> 4
                kk = 4 * sg + g
                if kk >= KGY:                           # half stage: the lanes past the last k group re-read groups 16 / 17 (their weight fragment is zero)
                    kk = KGY - 2 + (g & 1)
                return (kk // CG8 + li + 16 * qq) * PSTR + (kk % CG8) * 16
            assert worst_conflict(rd, B128_READ_GROUPS, 64) == 1, (sg, qq)


def producer_item(ptid):
    if ptid < 160:
        return ptid // 40, (ptid >> 1) & 3, (ptid & 1) + 2 * ((ptid >> 3) % 5)
    u = ptid - 160
    return ((u >> 2) & 1) + 2 * (u // 40), 4 + ((u >> 1) & 1), (u & 1) + 2 * ((u >> 3) % 5)


def test_producer_items_cover_the_block_once_and_pair_lanes():
    seen = set()
    for ptid in range(240):
        row, cig, q = producer_item(ptid)
        assert 0 <= row < 4 and 0 <= cig < 6 and 0 <= q < 10
        seen.add((row, cig, q))
    assert len(seen) == 240 == 4 * 6 * 10
    for ptid in range(0, 240, 2):                      # lane pairs: the same row and channel group, adjacent quads (32 contiguous bytes per plane)
        a, b = producer_item(ptid), producer_item(ptid + 1)
        assert a[:2] == b[:2] and b[2] == a[2] + 1 and a[2] % 2 == 0


def test_producer_writes_are_at_most_two_way():
    # pixel i of the quad: window column 4 q - 3 + i, written when it is inside 0..33
    for wave in range(4):
        for i in range(4):
            def wr(lane, wave=wave, i=i):
                ptid = wave * 64 + lane
                if ptid >= 240:
                    return None
                row, cig, q = producer_item(ptid)
                col = 4 * q - 3 + i
                if not 0 <= col < IW:
                    return None
                return (row * IW + col) * PSTR + cig * 16
            assert worst_conflict(wr, B128_WRITE_GROUPS, 32) <= 2, (wave, i)
    # the naive order (quads fastest over eight lanes) would be 8-way
    naive = lambda lane: (0 * IW + 4 * (lane & 7) - 3 + 3) * PSTR if lane < 8 else None
    assert worst_conflict(naive, B128_WRITE_GROUPS, 32) == 8


def _compile_pack(tmp_path):
    src = open(HDR).read()
    a, b = src.index("struct Conv3hCfg {"), src.index("// Block walk.")
    prog = ("#include <cstdint>\n#include <cstring>\n#include <cmath>\n#include <cstdio>\n#include <vector>\n#include <algorithm>\n" + src[a:b] +
            "int main() { std::vector<float> w(48 * 48 * 9); if (fread(w.data(), 4, w.size(), stdin) != w.size()) return 1; std::vector<uint32_t> img; "
            "conv3h_pack(w.data(), img); fwrite(img.data(), 4, img.size(), stdout); return 0; }\n")
    cpp, exe = tmp_path / "pack3h.cpp", tmp_path / "pack3h"
    cpp.write_text(prog)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), str(cpp)])
    return exe


def unpack_image(img):
    """image (uint32 words) -> h, l [48 co, 48 ci, 3, 3] float64 in scaled units, ex [48]"""
    b = img.view(np.uint8)
    ex = img[3 * WKY // 4: 3 * WKY // 4 + 48].astype(np.int32)
    h = np.full((C, C, 3, 3), np.nan)
    l = np.full((C, C, 3, 3), np.nan)
    for ky in range(3):
        for sg in range(SPK):
            half = sg == SPK - 1
            for n in range(3):
                for lane in range(32 if half else 64):
                    co, kk = n * 16 + (lane & 15), 4 * sg + (lane >> 4)
                    kx, c0 = kk // CG8, (kk % CG8) * 8
                    step = 512 if half else 1024
                    fb = ky * WKY + (4 * 6144 + n * 2 * 512 if half else sg * 6144 + n * 2 * 1024)
                    hh = np.frombuffer(b[fb + lane * 16: fb + lane * 16 + 16].tobytes(), np.float16).astype(np.float64)
                    ll = np.frombuffer(b[fb + step + lane * 16: fb + step + lane * 16 + 16].tobytes(), np.float16).astype(np.float64)
                    assert np.isnan(h[co, c0:c0 + 8, ky, kx]).all(), "two fragment slots hold the same weight"
                    h[co, c0:c0 + 8, ky, kx] = hh
                    l[co, c0:c0 + 8, ky, kx] = ll
    assert not np.isnan(h).any() and not np.isnan(l).any(), "a weight has no fragment slot"
    return h, l, ex


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_weight_image(tmp_path):
    exe = _compile_pack(tmp_path)
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((C, C, 3, 3)) * 10.0 ** rng.uniform(-3, 2, size=(C, 1, 1, 1))).astype(np.float32)
    w[5] = 0                                            # an all-zero output channel
    out = subprocess.run([str(exe)], input=w.tobytes(), capture_output=True, check=True).stdout
    img = np.frombuffer(out, np.uint32)
    assert img.size == 3 * WKY // 4 + 48
    h, l, ex = unpack_image(img)
    for co in range(C):
        m = np.abs(w[co]).max()
        if m == 0:
            assert ex[co] == 0 and not h[co].any() and not l[co].any()
            continue
        assert 2.0 ** 14 <= m * 2.0 ** int(ex[co]) < 2.0 ** 15           # one exponent per OUTPUT channel, its largest weight in [2^14, 2^15)
        ws = w[co].astype(np.float64) * 2.0 ** int(ex[co])
        assert np.array_equal(h[co], ws.astype(np.float32).astype(np.float16).astype(np.float64))      # h = RNE_f16(w 2^e)
        assert np.abs(h[co] + l[co] - ws).max() <= 2.0 ** -22 * 2.0 ** 15                               # two parts: 22 bits of the scaled range


def f16(x):
    return x.astype(np.float32).astype(np.float16).astype(np.float64)


def emulate(x, w, bias, relu):
    """conv3h_kernel's arithmetic for one strip walk in numpy: x [48, T, F], F <= 32 (one strip).  Returns y [48, T, F] and the exponents used."""
    T, F = x.shape[1:]
    tilesT = (T + 3) // 4
    ew = np.zeros(C, np.int64)
    for co in range(C):
        m = np.abs(w[co]).max()
        ew[co] = 15 - np.frexp(m)[1] if m > 0 else 0
    ws = w.astype(np.float64) * 2.0 ** ew[:, None, None, None]
    wh = f16(ws)
    wl = f16(ws - wh)
    xp = np.zeros((C, 4 * (tilesT + 1) + 4, F + 2))                       # rows -3 .. : block j = rows 4 j + 1 .. 4 j + 4 -> index 4 j + 4 ..
    xp[:, 3:3 + T, 1:1 + F] = x
    xh, xl, eb = np.zeros_like(xp), np.zeros_like(xp), []
    e_prev = 0
    for j in range(-1, tilesT):
        blk = xp[:, 4 * j + 4: 4 * j + 8]
        m = np.float32(np.abs(blk).max())
        need = min((15 - np.frexp(m)[1] if m > 0 else 15) - 1, 100)
        e = e_prev
        if j < 0 or need < e_prev:
            e = need
        elif need > e_prev + 8:
            e = min(need, e_prev + 40)
        e_prev = e
        eb.append(e)
        s = blk * 2.0 ** e
        xh[:, 4 * j + 4: 4 * j + 8] = f16(s)
        xl[:, 4 * j + 4: 4 * j + 8] = f16(s - xh[:, 4 * j + 4: 4 * j + 8])
    y = np.zeros((C, T, F))
    for j in range(tilesT):
        for r in range(4):
            t = 4 * j + r
            if t >= T:
                continue
            acc = np.zeros((C, F), np.float32)
            e_at = None
            for ky in range(3):
                row = t + ky - 1 + 3                    # index into xp
                blk_of_row = (row - 4) // 4 + 1 if row >= 4 else 0          # position in eb (block -1 is eb[0])
                e_row = eb[blk_of_row]
                if e_at is not None and e_row != e_at:
                    acc = np.ldexp(acc, e_row - e_at).astype(np.float32)
                e_at = e_row
                for kx in range(3):
                    a_h, a_l = xh[:, row, kx:kx + F], xl[:, row, kx:kx + F]
                    # w_l x_h + w_h x_l + w_h x_h, products exact, fp32 accumulation (the MFMA's order inside a stage is not modelled: float64 sum, one rounding)
                    acc = (acc.astype(np.float64) + wl[:, :, ky, kx] @ a_h + wh[:, :, ky, kx] @ a_l + wh[:, :, ky, kx] @ a_h).astype(np.float32)
            out = acc.astype(np.float64) * 2.0 ** (-(e_at + ew))[:, None] + bias[:, None]
            y[:, t] = np.maximum(out, 0) if relu else out
    return y, eb


@pytest.mark.parametrize("T,spread,relu", [(12, 0.0, True), (37, 3.0, False), (67, 8.0, False), (21, -1.0, False)])
def test_arithmetic_emulation_vs_float64(T, spread, relu):
    rng = np.random.default_rng(T)
    F = 32
    x = rng.standard_normal((C, T, F)).astype(np.float32)
    if spread >= 0:
        x *= (10.0 ** (spread * np.cos(0.21 * np.arange(T))))[None, :, None].astype(np.float32)
    else:
        # a loud passage (1e15), two all-zero blocks, then unit-size data: the exponent has to rise by more than 2^40 -- the bounded-rise rule
        x[:, :8] *= np.float32(1e15)
        x[:, 8:16] = 0
    w = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C) * 10.0 ** rng.uniform(-1, 1, size=(C, 1, 1, 1))).astype(np.float32)
    bias = rng.standard_normal(C).astype(np.float32)
    y, eb = emulate(x, w, bias.astype(np.float64), relu)
    assert len(set(eb)) > 1 or spread == 0, "the exponent never moved: the rescale path is untested"
    if spread < 0:
        d = np.diff(eb)
        assert d.max() == 40, "the bounded rise (a block more than 2^40 quieter) did not occur"
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1)))
    ref = np.zeros((C, T, F))
    mag = np.zeros((C, T, F))
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum("oc,ctf->otf", w[:, :, ky, kx].astype(np.float64), xp[:, ky:ky + T, kx:kx + F])
            mag += np.einsum("oc,ctf->otf", np.abs(w[:, :, ky, kx]).astype(np.float64), np.abs(xp[:, ky:ky + T, kx:kx + F]))
    ref += bias[:, None, None]
    mag += np.abs(bias)[:, None, None]
    if relu:
        ref = np.maximum(ref, 0)
    # fp32-grade: relative to sum |w x| (what an fp32 chain's roundings are relative to)
    assert (np.abs(y - ref) <= 2e-6 * mag + 1e-300).all(), float((np.abs(y - ref) / (mag + 1e-300)).max())
