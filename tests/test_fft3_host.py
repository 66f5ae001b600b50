"""The per-thread stage bodies of the fast FFT path (csrc/kernels_fft3.h: radix-12 / 16 / 16 Stockham passes, padded LDS
exchange layout, forward split, inverse merge) compiled for the HOST (g++, ASX_HOST_TEST) and run thread by thread by
tests/host/fft3_host.cpp, against numpy's FFT.  The same source lines run in stft3_kernel / istft3_kernel on the GPU
(tests/test_gpu_parity.py covers those through the C ABI); this test pins the index maps and twiddles without a GPU."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fft3_passes_match_numpy(tmp_path):
    exe = str(tmp_path / "fft3_host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host", "fft3_host.cpp")], check=True)
    rng = np.random.default_rng(5)
    n = 6144
    win = 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))
    x = (rng.standard_normal(n) * win).astype(np.float32)
    X = (rng.standard_normal(3072) + 1j * rng.standard_normal(3072)).astype(np.complex64)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(x.tobytes())
        f.write(np.stack([X.real, X.imag], 1).astype(np.float32).tobytes())
    subprocess.run([exe, fin, fout], check=True)
    out = np.fromfile(fout, np.float32)
    fwd = out[: 2 * 3072].reshape(3072, 2)
    fwd = fwd[:, 0] + 1j * fwd[:, 1]
    ref = np.fft.rfft(x.astype(np.float64))[:3072]
    err = np.sqrt(np.mean(np.abs(fwd - ref) ** 2) / np.mean(np.abs(ref) ** 2))
    assert err < 2e-6, err
    inv = out[2 * 3072:]
    Xf = np.zeros(3073, np.complex128)
    Xf[:3072] = X
    Xf[0] = Xf[0].real                      # c2r ignores the imaginary part of DC; the Nyquist bin is absent (dim_f = 3072)
    refi = np.fft.irfft(Xf, n)
    erri = np.sqrt(np.mean((inv - refi) ** 2) / np.mean(refi ** 2))
    assert erri < 2e-6, erri
