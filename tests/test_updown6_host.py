"""CPU checks of the level-change kernels (csrc/kernels_updown6.h: conv_down6_kernel / conv_up6_kernel, bf16 x 6):
  * the host weight packers (cut out of the header, compiled with g++): every weight lands in the fragment slot the kernel reads it from, the three
    bf16 parts sum to the fp32 weight EXACTLY, channel groups / stages / virtual tiles past the layer are zero;
  * the LDS layout claims: the x fragment reads of both kernels are conflict-free on gfx950's 64 banks (`ds_read_b64`: 32 lanes per pass;
    `ds_read_b32`: all 64 lanes), every DMA piece is one contiguous KB;
  * a numpy emulation of the arithmetic (exact three-way bf16 split, six products, fp32 accumulation) against float64."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "python-audio-separator_amd", "csrc", "kernels_updown6.h")


def _compile(tmp_path):
    src = open(HDR).read()
    a = src.index("// NREP: 16-channel tiles of the output per workgroup")
    b = src.index("// the six products of one 32-deep k step")
    c = src.index("template <int NREP_>\nstruct Up6CfgT {")
    d = src.index("// a.T / a.F: INPUT plane; output [B, Cout, 2 T, 2 F]")
    prog = ("#include <cstdint>\n#include <cstring>\n#include <cmath>\n#include <cstdio>\n#include <cstdlib>\n#include <vector>\n" + src[a:b] + src[c:d] +
            "int main(int argc, char **argv) { const int kind = atoi(argv[1]), nrep = atoi(argv[2]), cout = atoi(argv[3]), cin = atoi(argv[4]);\n"
            "  std::vector<float> w((size_t)cout * cin * 4); if (fread(w.data(), 4, w.size(), stdin) != w.size()) return 1;\n"
            "  std::vector<uint32_t> img; int cg = 0, nst = 0;\n"
            "  if (kind == 0) { if (nrep == 6) down6_pack<6>(w.data(), cout, cin, img, &cg, &nst); else down6_pack<3>(w.data(), cout, cin, img, &cg, &nst); }\n"
            "  else { if (nrep == 6) up6_pack<6>(w.data(), cout, cin, img, &cg, &nst); else if (nrep == 4) up6_pack<4>(w.data(), cout, cin, img, &cg, &nst);\n"
            "         else up6_pack<2>(w.data(), cout, cin, img, &cg, &nst); }\n"
            "  uint32_t hdr[2] = {(uint32_t)cg, (uint32_t)nst}; fwrite(hdr, 4, 2, stdout); fwrite(img.data(), 4, img.size(), stdout); return 0; }\n")
    cpp, exe = tmp_path / "pack6.cpp", tmp_path / "pack6"
    cpp.write_text(prog)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), str(cpp)])
    return exe


def bf16_to_f64(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def _run(exe, kind, nrep, w, cout, cin):
    out = subprocess.run([str(exe), str(kind), str(nrep), str(cout), str(cin)], input=w.astype(np.float32).tobytes(), capture_output=True, check=True).stdout
    cg, nst = np.frombuffer(out[:8], np.uint32)
    img = np.frombuffer(out[8:], np.uint16)
    return int(cg), int(nst), img.reshape(int(cg), int(nst), 3, nrep, 64, 8)


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
@pytest.mark.parametrize("nrep,cout,cin", [(3, 96, 48), (6, 96, 48), (3, 144, 96), (3, 40, 20), (6, 192, 50)])
def test_down_weight_image(tmp_path, nrep, cout, cin):
    exe = _compile(tmp_path)
    rng = np.random.default_rng(cout + cin)
    w = (rng.standard_normal((cout, cin, 2, 2)) * 10.0 ** rng.uniform(-4, 3, size=(cout, 1, 1, 1))).astype(np.float32)
    cg, nst, img = _run(exe, 0, nrep, w, cout, cin)
    assert cg == -(-cout // (16 * nrep)) and nst == -(-cin // 8)
    parts = bf16_to_f64(img)                              # [cg, st, part, n, lane, 8]
    total = parts.sum(axis=2)
    for g in range(cg):
        for st in range(nst):
            for n in range(nrep):
                for lane in range(64):
                    co, lk = g * 16 * nrep + n * 16 + (lane & 15), lane >> 4
                    for e in range(2):
                        c = st * 8 + 2 * lk + e
                        want = w[co, c].reshape(4).astype(np.float64) if (co < cout and c < cin) else np.zeros(4)
                        got = total[g, st, n, lane, e * 4: e * 4 + 4]              # k = e * 4 + dy * 2 + dx
                        assert np.array_equal(got, want), (g, st, n, lane, e)       # h + m + l == w EXACTLY
    # h is the round-to-nearest-even bf16 of w
    co, c = 3, 5
    h = parts[0, 0, 0, 0, (c // 2) * 16 + co, (c & 1) * 4 + 3]
    u = np.float32(w[co, c, 1, 1]).view(np.uint32)
    assert h == bf16_to_f64(np.array([(u + 0x7fff + ((u >> 16) & 1)) >> 16], np.uint32).astype(np.uint16))[0]


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
@pytest.mark.parametrize("nrep,cout,cin", [(6, 48, 96), (6, 96, 144), (4, 32, 64), (2, 20, 100), (2, 8, 16)])
def test_up_weight_image(tmp_path, nrep, cout, cin):
    exe = _compile(tmp_path)
    rng = np.random.default_rng(cout * 3 + cin)
    w = (rng.standard_normal((cin, cout, 2, 2)) * 10.0 ** rng.uniform(-4, 3, size=(1, cout, 1, 1))).astype(np.float32)
    cg, nst, img = _run(exe, 1, nrep, w, cout, cin)
    CT = -(-cout // 16)
    assert cg == -(-4 * CT // nrep) and nst == -(-cin // 32)
    total = bf16_to_f64(img).sum(axis=2)                  # [cg, st, n, lane, 8]
    for g in range(cg):
        for st in range(nst):
            for n in range(nrep):
                nt = g * nrep + n
                pair, dx = nt // 2, nt & 1
                dy, ct = pair // CT, pair % CT
                for lane in range(64):
                    co, lk = ct * 16 + (lane & 15), lane >> 4
                    for kk in range(8):
                        c = st * 32 + 2 * (lk + 4 * (kk >> 1)) + (kk & 1)
                        want = float(w[c, co, dy, dx]) if (dy < 2 and co < cout and c < cin) else 0.0
                        assert total[g, st, n, lane, kk] == want, (g, st, n, lane, kk)


def test_x_fragment_reads_are_conflict_free():
    # conv_down6_kernel: ds_read_b64 at float index (2 lk + e) * PS + (2 row + dy) * 128 + 2 * (col * 16 + li); 32 lanes per pass over 64 banks of 4 bytes
    PS = 4 * 128 + 16
    for e in range(2):
        for dy in range(2):
            for half in range(2):
                banks = set()
                for lane in range(32 * half, 32 * half + 32):
                    li, lk = lane & 15, lane >> 4
                    a = (2 * lk + e) * PS + dy * 128 + 2 * li
                    for wd in range(2):
                        b = (a + wd) % 64
                        assert b not in banks, (e, dy, half, lane)
                        banks.add(b)
    # conv_up6_kernel: ds_read_b32 at (lk + 4 jj) * PAIR + e * 128 + row * 64 + col * 16 + li; 64 lanes over 64 banks
    PAIR = 2 * 128 + 16
    for jj in range(4):
        for e in range(2):
            banks = set()
            for lane in range(64):
                li, lk = lane & 15, lane >> 4
                b = ((lk + 4 * jj) * PAIR + e * 128 + li) % 64
                assert b not in banks, (jj, e, lane)
                banks.add(b)
    # every DMA piece is one contiguous KB: a down plane is 2 (NJ) pieces at plane * PS * 4 + j * 1024; an up piece is a plane PAIR at pair * PAIR * 4
    assert PS * 4 % 16 == 0 and PAIR * 4 % 16 == 0 and PS * 4 >= 2 * 1024 and PAIR * 4 >= 1024


def _split3(v):
    """exact three-way bf16 split of float32 values (round to nearest even), as float64 parts"""
    def rne(x):
        u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7fff + ((u >> 16) & 1)) >> 16
        return (u.astype(np.uint32) << 16).view(np.float32)
    h = rne(v)
    r = (v.astype(np.float32) - h).astype(np.float32)
    m = rne(r)
    l = rne((r - m).astype(np.float32))
    return h.astype(np.float64), m.astype(np.float64), l.astype(np.float64)


def test_arithmetic_emulation_vs_float64():
    """six bf16 products per multiply-add (w_l x_h, w_m x_m, w_m x_h, w_h x_l, w_h x_m, w_h x_h; the three dropped ones <= 2^-24 of a product), fp32
    accumulation per 32-deep step: a stride-2 conv row against float64, channels 2^12 apart in magnitude"""
    rng = np.random.default_rng(7)
    cin, cout, npx = 96, 48, 64
    x = (rng.standard_normal((npx, 4 * cin)) * np.repeat(np.exp2(rng.integers(-6, 7, cin)), 4)).astype(np.float32)   # [pixel, k = (channel, tap)]
    w = (rng.standard_normal((cout, 4 * cin)) / np.sqrt(4 * cin)).astype(np.float32)
    xh, xm, xl = _split3(x)
    wh, wm, wl = _split3(w)
    assert np.array_equal(xh + xm + xl, x.astype(np.float64)) and np.array_equal(wh + wm + wl, w.astype(np.float64))
    acc = np.zeros((npx, cout), np.float32)
    for k0 in range(0, 4 * cin, 32):
        s = slice(k0, k0 + 32)
        for a, b in ((xh, wl), (xm, wm), (xh, wm), (xl, wh), (xm, wh), (xh, wh)):
            acc = (acc.astype(np.float64) + a[:, s] @ b[:, s].T).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    acc32 = np.zeros((npx, cout), np.float32)
    for k in range(4 * cin):                                # a plain fp32 fma chain for scale
        acc32 = (acc32.astype(np.float64) + np.outer(x[:, k].astype(np.float64), w[:, k].astype(np.float64))).astype(np.float32)
    e6 = np.sqrt(((acc - ref) ** 2).mean() / (ref ** 2).mean())
    e32 = np.sqrt(((acc32 - ref) ** 2).mean() / (ref ** 2).mean())
    mag = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T
    assert (np.abs(acc - ref) <= 4e-7 * mag).all()
    assert e6 <= e32, (e6, e32)                             # fewer accumulator roundings than the fma chain
