#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference).  The reference's
hot-path modules are imported unmodified; third-party packages that are absent
here and are never touched by ``demix``/``run_model``/``STFT`` (onnx,
onnxruntime, onnx2torch, librosa, soundfile, audioread, pydub,
pytorch_lightning) are satisfied with empty stub modules.

    python tests/golden/make_golden.py

Writes: stft_small.npz, stft_hq3_subsample.npz, net_small.npz, demix_small.npz,
stems_small.npz.  Inputs are regenerated from seeds by the tests (numpy
Generator streams are stable across versions); outputs are stored.
"""
import importlib.machinery
import logging
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    for name in ["onnx", "onnxruntime", "onnx2torch", "librosa", "soundfile", "audioread"]:
        _stub(name)
    _stub("pydub", AudioSegment=object)
    _stub("pytorch_lightning", LightningModule=torch.nn.Module)
    # bypass audio_separator/__init__.py (it imports the whole orchestrator)
    pkg = types.ModuleType("audio_separator")
    pkg.__path__ = [os.path.join(REF, "audio_separator")]
    pkg.__spec__ = importlib.machinery.ModuleSpec("audio_separator", None, is_package=True)
    sys.modules["audio_separator"] = pkg
    sep = types.ModuleType("audio_separator.separator")
    sep.__path__ = [os.path.join(REF, "audio_separator", "separator")]
    sep.__spec__ = importlib.machinery.ModuleSpec("audio_separator.separator", None, is_package=True)
    sys.modules["audio_separator.separator"] = sep
    from audio_separator.separator.architectures.mdx_separator import MDXSeparator
    from audio_separator.separator.uvr_lib_v5.stft import STFT
    from audio_separator.separator.uvr_lib_v5.mdxnet import ConvTDFNet
    from audio_separator.separator.uvr_lib_v5 import spec_utils
    return MDXSeparator, STFT, ConvTDFNet, spec_utils


def make_ref_separator(MDXSeparator, n_fft, hop, dim_f, segment, overlap, denoise, model_run):
    s = MDXSeparator.__new__(MDXSeparator)
    s.logger = logging.getLogger("golden")
    s.n_fft, s.hop_length, s.dim_f = n_fft, hop, dim_f
    s.segment_size, s.overlap, s.batch_size = segment, overlap, 1
    s.enable_denoise = denoise
    s.torch_device = torch.device("cpu")
    s.model_run = model_run
    return s


def main():
    sys.path.insert(0, ROOT)
    from oracle import mdx_oracle as O
    MDXSeparator, STFT, ConvTDFNet, spec_utils = import_reference()
    log = logging.getLogger("golden")
    torch.set_num_threads(os.cpu_count())

    # 1. STFT small (n_fft = 96 = 3 * 2^5, same radix structure as 6144 = 3 * 2^11)
    n_fft, hop, dim_f = 96, 16, 40
    rng = np.random.default_rng(11)
    x = (0.5 * rng.standard_normal((2, 2, hop * 15))).astype(np.float32)
    st = STFT(log, n_fft, hop, dim_f, torch.device("cpu"))
    X = st(torch.tensor(x)).numpy()
    rngs = np.random.default_rng(12)
    S = rngs.standard_normal(X.shape).astype(np.float32)
    y = st.inverse(torch.tensor(S)).numpy()
    np.savez_compressed(os.path.join(HERE, "stft_small.npz"), n_fft=n_fft, hop=hop, dim_f=dim_f,
                        x_seed=11, s_seed=12, X=X, y=y)

    # 2. STFT at HQ_3 geometry, outputs subsampled to keep the fixture small
    n_fft, hop, dim_f, seg = 6144, 1024, 3072, 256
    C = hop * (seg - 1)
    rng = np.random.default_rng(21)
    x = (0.3 * rng.standard_normal((1, 2, C))).astype(np.float32)
    st = STFT(log, n_fft, hop, dim_f, torch.device("cpu"))
    X = st(torch.tensor(x)).numpy()
    fsel = np.arange(0, dim_f, 61)
    tsel = np.arange(0, seg, 15)
    rngs = np.random.default_rng(22)
    S = rngs.standard_normal((1, 4, dim_f, seg)).astype(np.float32)
    y = st.inverse(torch.tensor(S)).numpy()
    ysel = np.concatenate([np.arange(0, 4096), np.arange(4096, C - 4096, 509), np.arange(C - 4096, C)])
    np.savez_compressed(os.path.join(HERE, "stft_hq3_subsample.npz"), n_fft=n_fft, hop=hop, dim_f=dim_f, seg=seg,
                        x_seed=21, s_seed=22, fsel=fsel, tsel=tsel, X_sub=X[:, :, fsel][:, :, :, tsel],
                        ysel=ysel, y_sub=y[:, :, ysel],
                        X_abs_sum=np.float64(np.abs(X.astype(np.float64)).sum()),
                        y_abs_sum=np.float64(np.abs(y.astype(np.float64)).sum()))

    # 3. ConvTDFNet small: the reference class with the oracle's synthetic state_dict
    d = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4, bias=False)
    sd = O.make_convtdf_state(d, seed=3)
    net = ConvTDFNet(target_name="t", lr=1e-3, optimizer="rmsprop", dim_c=d.dim_c, dim_f=d.dim_f, dim_t=d.dim_t,
                     n_fft=96, hop_length=16, num_blocks=d.num_blocks, l=d.l, g=d.g, k=d.k, bn=d.bn, bias=d.bias,
                     overlap=0)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(m in ("window", "freq_pad") or m.endswith("num_batches_tracked") for m in missing), missing
    net.eval()
    rng = np.random.default_rng(31)
    xin = rng.standard_normal((2, 4, d.dim_f, d.dim_t)).astype(np.float32)
    with torch.no_grad():
        yout = net(torch.tensor(xin)).numpy()
    np.savez_compressed(os.path.join(HERE, "net_small.npz"), dims=np.array([d.dim_c, d.dim_f, d.dim_t, d.g, d.l, d.num_blocks, d.k, d.bn]),
                        w_seed=3, x_seed=31, y=yout)

    # also with TDF bias=True
    d2 = O.NetDims(dim_c=4, dim_f=32, dim_t=16, g=8, l=2, num_blocks=5, k=3, bn=4, bias=True)
    sd2 = O.make_convtdf_state(d2, seed=4)
    net2 = ConvTDFNet("t", 1e-3, "rmsprop", 4, 32, 16, 96, 16, 5, 2, 8, 3, 4, True, 0)
    net2.load_state_dict(sd2, strict=False)
    net2.eval()
    with torch.no_grad():
        yout2 = net2(torch.tensor(xin)).numpy()
    np.savez_compressed(os.path.join(HERE, "net_small_bias.npz"), w_seed=4, x_seed=31, y=yout2)

    # 4. demix small: reference chunk loop + reference net
    def model_run(spek):
        with torch.no_grad():
            return net(torch.as_tensor(spek, dtype=torch.float32))
    cases = {}
    N = 3000
    rng = np.random.default_rng(41)
    mix = (0.4 * rng.standard_normal((2, N))).astype(np.float32)
    for name, overlap, denoise, match in [("ov25", 0.25, False, False), ("ov25_denoise", 0.25, True, False),
                                          ("ov0", 0.0, False, False), ("ov75", 0.75, False, False),
                                          ("match", 0.25, False, True)]:
        s = make_ref_separator(MDXSeparator, 96, 16, 32, 16, overlap, denoise, model_run)
        cases[name] = s.demix(mix.copy(), is_match_mix=match).astype(np.float32)
    # ragged / tiny inputs
    for name, n in [("n1", 1), ("n143", 143), ("n144", 144), ("n145", 145)]:
        rngn = np.random.default_rng(100 + n)
        mixn = (0.4 * rngn.standard_normal((2, n))).astype(np.float32)
        s = make_ref_separator(MDXSeparator, 96, 16, 32, 16, 0.25, False, model_run)
        cases["ragged_" + name] = s.demix(mixn, is_match_mix=False).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "demix_small.npz"), mix_seed=41, N=N, **cases)

    # 5. stem algebra (mdx_separator.py:155-182), loud input so normalize() scales
    rng = np.random.default_rng(51)
    mix = (0.8 * rng.standard_normal((2, 2000))).astype(np.float32)
    s = make_ref_separator(MDXSeparator, 96, 16, 32, 16, 0.25, False, model_run)
    s.compensate = 1.035
    m = mix.copy()
    peak = np.abs(m).max()
    m = spec_utils.normalize(wave=m, max_peak=0.9, min_peak=0.0)
    source = s.demix(m) * peak
    primary = source.T
    secondary = (-primary * s.compensate) + m.T
    np.savez_compressed(os.path.join(HERE, "stems_small.npz"), mix_seed=51, compensate=1.035,
                        primary=primary.astype(np.float32), secondary=secondary.astype(np.float32), mix_norm=m)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
