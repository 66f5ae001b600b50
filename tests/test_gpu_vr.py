"""GPU parity of the VR path against golden vectors written by the reference's spec_utils functions / nets.py classes
(tests/golden/make_golden_vr.py) and against the CPU oracle.  Bars: 1e-4 relative RMS on separated waves and masks,
2e-5 on the analysis spectrogram."""
import os

import numpy as np
import pytest

from oracle import vr_oracle as V

pytestmark = pytest.mark.gpu
TOL = 1e-4
SMALL_CAP = [(2, 4), (2, 4), (6, 4, 1, 1, 0), (4, 4), (10, 4, 1, 1, 0), (4, 8), (8, 2, 1), (4, 2, 1), (4, 2, 1)]


def rel_rms(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2)) / max(np.sqrt(np.mean(np.abs(b) ** 2)), 1e-30))


@pytest.fixture(scope="module")
def A():
    import audio_separator_amd as A
    return A


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "vr_small.npz"))


def demixer(A, arch=123821, seed=5, params=None, max_batch=0, **arch_cfg):
    mp = params or V.small_params().param
    # the goldens were written through a polyphase stand-in for librosa.resample: pin the converter (the class default is the
    # reference's platform rule, sinc_fastest on Linux)
    cfg = {"window_size": 64, "batch_size": 2, "aggression": 5, "asx_res_type": "polyphase"}
    cfg.update(arch_cfg)
    return A.VRDemixer({"model_params": mp, "primary_stem_name": "Instrumental", "torch_device": 0}, cfg,
                       state_dict=V.make_vr_state(arch, seed, SMALL_CAP), nn_arch_size=arch, capacity=SMALL_CAP, offset=16,
                       max_batch=max_batch)


@pytest.mark.parametrize("tag,arch,seed", [("hp", 123821, 5), ("sp7", 33966, 6)])
def test_net_golden(A, g, tag, arch, seed):
    dm = demixer(A, arch, seed)
    y = dm.engine.vr_forward(g[f"{tag}_net_in"])
    assert rel_rms(y, g[f"{tag}_net_out"]) < TOL, rel_rms(y, g[f"{tag}_net_out"])


def test_analysis_golden(A, g):
    dm = demixer(A)
    X = dm.engine.vr_analysis(g["wave"])
    assert X.shape == g["X_spec"].shape
    assert rel_rms(X, g["X_spec"]) < 2e-5, rel_rms(X, g["X_spec"])
    pm = dict(V.small_params().param)
    pm["mid_side"] = True
    Xm = demixer(A, params=pm).engine.vr_analysis(g["wave"])
    assert rel_rms(Xm, g["ms_X_spec"]) < 2e-5, rel_rms(Xm, g["ms_X_spec"])


@pytest.mark.parametrize("max_batch", [0, 3])
def test_separate_golden(A, g, max_batch):
    p, s = demixer(A, max_batch=max_batch).separate_stems(g["wave"])
    assert p.shape == g["wav_y"].T.shape
    assert rel_rms(p, g["wav_y"].T) < TOL, rel_rms(p, g["wav_y"].T)
    assert rel_rms(s, g["wav_v"].T) < TOL, rel_rms(s, g["wav_v"].T)


@pytest.mark.parametrize("kw", [dict(enable_tta=True), dict(enable_post_process=True), dict(aggression=0),
                                dict(aggression=20, enable_tta=True, enable_post_process=True, post_process_threshold=0.1)])
def test_separate_options_oracle(A, g, kw):
    """TTA / merge_artifacts / aggression variants against the oracle (itself pinned on the reference for each of them)"""
    mp = V.small_params()
    sd = V.make_vr_state(123821, 5, SMALL_CAP)
    okw = dict(window_size=64, batch_size=2, aggression=kw.get("aggression", 5), enable_tta=kw.get("enable_tta", False),
               enable_post_process=kw.get("enable_post_process", False), post_process_threshold=kw.get("post_process_threshold", 0.2),
               offset=16)
    wp, ws = V.vr_separate(g["wave"], sd, 123821, mp, **okw)
    p, s = demixer(A, **kw).separate_stems(g["wave"])
    assert rel_rms(p, wp) < TOL, rel_rms(p, wp)
    assert rel_rms(s, ws) < TOL, rel_rms(s, ws)


def test_mid_side_and_non_accom(A, g):
    pm = dict(V.small_params().param)
    pm["mid_side"] = True
    pm["aggr_correction"] = {"left": 0.02, "right": -0.03}
    mp = V.ModelParams(pm)
    sd = V.make_vr_state(123821, 5, SMALL_CAP)
    wave = g["wave"][:, :9001]
    wp, ws = V.vr_separate(wave, sd, 123821, mp, window_size=64, batch_size=1, aggression=10, is_non_accom_stem=True, offset=16)
    dm = A.VRDemixer({"model_params": pm, "primary_stem_name": "Vocals", "torch_device": 0},
                     {"window_size": 64, "batch_size": 1, "aggression": 10, "asx_res_type": "polyphase"}, state_dict=sd, nn_arch_size=123821,
                     capacity=SMALL_CAP, offset=16)
    p, s = dm.separate_stems(wave)
    assert rel_rms(p, wp) < TOL, rel_rms(p, wp)
    assert rel_rms(s, ws) < TOL, rel_rms(s, ws)


# ---- sinc_fastest: libsamplerate's converter (restated; parity unpinned: oracle/vr_oracle.py src_simple_sinc_fastest) --------------
def test_default_converter_follows_the_reference_platform_rule(A):
    """spec_utils.py:33-38: sinc_fastest everywhere but macOS on ARM"""
    import platform
    from audio_separator_amd import vr
    want = "polyphase" if (platform.system() == "Darwin" and "arm" in (platform.processor() + platform.platform()).lower()) else "sinc_fastest"
    assert vr.resolve_res_type(None) == want == vr.resolve_res_type("auto")
    assert vr.resolve_res_type("sinc") == "sinc_fastest" and vr.resolve_res_type("polyphase") == "polyphase"
    cfg = {"window_size": 64, "batch_size": 2, "aggression": 5}
    dm = A.VRDemixer({"model_params": V.small_params().param, "primary_stem_name": "Instrumental", "torch_device": 0}, cfg,
                     state_dict=V.make_vr_state(123821, 5, SMALL_CAP), nn_arch_size=123821, capacity=SMALL_CAP, offset=16)
    assert dm.wav_resolution == want
    with pytest.raises(ValueError):
        vr.resolve_res_type("kaiser_best")


@pytest.mark.parametrize("res", ["polyphase", "sinc_fastest"])
@pytest.mark.parametrize("kw", [dict(), dict(enable_tta=True, aggression=10), dict(high_end_process=True)])
def test_separate_both_converters_vs_oracle(A, g, res, kw):
    """Synthesis chain (band d -> d + 1, ratio 2) on either converter against the oracle's restatement of it"""
    mp = V.small_params()
    sd = V.make_vr_state(123821, 5, SMALL_CAP)
    wp, ws = V.vr_separate(g["wave"], sd, 123821, mp, window_size=64, batch_size=2, aggression=kw.get("aggression", 5),
                           enable_tta=kw.get("enable_tta", False), offset=16, high_end_process=kw.get("high_end_process", False),
                           wav_resolution=res)
    p, s = demixer(A, asx_res_type=res, **kw).separate_stems(g["wave"])
    assert p.shape == wp.shape
    assert rel_rms(p, wp) < TOL, rel_rms(p, wp)
    assert rel_rms(s, ws) < TOL, rel_rms(s, ws)
    if res == "sinc_fastest":       # and the two converters really are different chains (about 1e-3 apart on this layout)
        pp, _ = demixer(A, asx_res_type="polyphase", **kw).separate_stems(g["wave"])
        assert rel_rms(p, pp) > 1e-5


def two_band_params(ratio_lo, res_lo="sinc_fastest"):
    """2band_44100_lofi.json's structure scaled down: band 1 is reached by DOWN-sampling with libsamplerate (analysis) and left by
    up-sampling with it (synthesis); sr_lo / sr_hi = 1/4 (lofi), 3/16 (2band_32000) or 1/8 (2band_48000)."""
    sr_hi = 8000
    sr_lo = int(sr_hi * ratio_lo)
    hl_lo = {0.25: 16, 0.1875: 12, 0.125: 8}[ratio_lo]
    return V.ModelParams({
        "bins": 96, "unstable_bins": 2, "reduction_bins": 90,
        "band": {1: {"sr": sr_lo, "hl": hl_lo, "n_fft": 128, "crop_start": 0, "crop_stop": 36, "lpf_start": 10, "lpf_stop": 30, "res_type": res_lo},
                 2: {"sr": sr_hi, "hl": 64, "n_fft": 192, "crop_start": 4, "crop_stop": 64, "hpf_start": 10, "hpf_stop": 4, "res_type": "sinc_medium"}},
        "sr": sr_hi, "pre_filter_start": 94, "pre_filter_stop": 96})


@pytest.mark.parametrize("ratio_lo", [0.25, 0.1875, 0.125])
def test_two_band_sinc_analysis_and_synthesis_vs_oracle(A, g, ratio_lo):
    """Band 1 with res_type "sinc_fastest": the analysis chain down-samples through the converter (ratio < 1: the table is walked at
    128 * ratio entries per input sample, gain ratio), the synthesis chain up-samples by 4 / 16:3 / 8 (non-dyadic position recurrence)"""
    mp = two_band_params(ratio_lo)
    sd = V.make_vr_state(123821, 7, SMALL_CAP)
    wave = g["wave"][:, :16000]
    X = V.loading_mix(wave, mp)
    dm = demixer(A, seed=7, params=mp.param, asx_res_type="sinc_fastest")
    Xg = dm.engine.vr_analysis(wave)
    assert Xg.shape == X.shape
    assert rel_rms(Xg, X) < 2e-5, rel_rms(Xg, X)
    wp, ws = V.vr_separate(wave, sd, 123821, mp, window_size=64, batch_size=2, aggression=5, offset=16, wav_resolution="sinc_fastest")
    p, s = dm.separate_stems(wave)
    assert p.shape == wp.shape
    assert rel_rms(p, wp) < TOL, rel_rms(p, wp)
    assert rel_rms(s, ws) < TOL, rel_rms(s, ws)
    # the polyphase analysis of the same band is a different spectrogram: the band's res_type is honoured
    Xp = demixer(A, seed=7, params=two_band_params(ratio_lo, "polyphase").param, asx_res_type="sinc_fastest").engine.vr_analysis(wave)
    assert rel_rms(Xp, X) > 1e-5


def test_v51_sinc_vs_oracle(A, g):
    mp = V.small_params_v51()
    sd = V.make_vr51_state(192, 16, 16, 9)
    wp, ws = V.vr_separate_v51(g["wave"][:, :12001], sd, mp, window_size=64, batch_size=2, aggression=5, offset=16, wav_resolution="sinc_fastest")
    p, s = demixer51(A, asx_res_type="sinc_fastest").separate_stems(g["wave"][:, :12001])
    assert rel_rms(p, wp) < TOL, rel_rms(p, wp)
    assert rel_rms(s, ws) < TOL, rel_rms(s, ws)


@pytest.mark.parametrize("ratio", [2.0, 3.0, 16 / 3, 4.0, 8.0, 0.25, 0.1875, 0.125, 2 ** (2 / 12), 2 ** (-3 / 12), 44100 / 32000])
@pytest.mark.parametrize("mono", [False, True])
def test_resample_sinc_hook_vs_oracle(A, ratio, mono):
    """asx_resample_sinc -- the converter itself, outside the VR chain -- against the restated libsamplerate algorithm: rational
    up / down ratios of the shipped band layouts, the irrational ratios of the pitch-shift round trip, one stereo call or one
    call per channel (the library's end-of-input test drops the last frame of a mono call when n * ratio is an integer)"""
    eng = A.Engine(A.MDXConfig(n_fft=96, hop_length=16, dim_f=32, segment_size=16))
    rng = np.random.default_rng(3)
    for n in (1, 37, 4096, 30011):
        x = rng.standard_normal((2, n)).astype(np.float32)
        got = eng.resample_sinc(x, ratio, mono_calls=mono)
        ref = V.src_simple_sinc_fastest(x, float(ratio), mono=mono)
        n_out = int(np.ceil(n * float(ratio)))
        assert got.shape == (2, n_out)
        ref = np.pad(ref, ((0, 0), (0, n_out - ref.shape[1])))
        if ref.size and np.abs(ref).max() > 0:
            assert rel_rms(got, ref) < 2e-6, (n, rel_rms(got, ref))
        else:
            assert not got.any()
    eng.close()


def test_error_paths(A):
    dm = demixer(A)
    with pytest.raises(ValueError):
        dm.separate_stems(np.zeros((1, 1000), np.float32))
    with pytest.raises(A.AsxError):
        demixer(A, window_size=60)      # not a multiple of 16


# ---- VR 5.1 (nets_new.CascadedNet + is_v51_model branches) --------------------------------------------------------------
@pytest.fixture(scope="module")
def g51(golden_dir):
    return np.load(os.path.join(golden_dir, "vr51_small.npz"))


def demixer51(A, **arch_cfg):
    cfg = {"window_size": 64, "batch_size": 2, "aggression": 5, "asx_res_type": "polyphase"}
    cfg.update(arch_cfg)
    return A.VRDemixer({"model_params": V.small_params_v51().param, "primary_stem_name": "Instrumental", "torch_device": 0,
                        "model_data": {"nout": 16, "nout_lstm": 16}}, cfg, state_dict=V.make_vr51_state(192, 16, 16, 9),
                       nn_arch_size=56817, offset=16)


def test_v51_net_and_analysis_golden(A, g, g51):
    dm = demixer51(A)
    y = dm.engine.vr_forward(g51["net_in"])
    assert rel_rms(y, g51["net_out"]) < TOL, rel_rms(y, g51["net_out"])
    X = dm.engine.vr_analysis(g["wave"])
    assert rel_rms(X, g51["X_spec"]) < 2e-5, rel_rms(X, g51["X_spec"])


def test_v51_separate_golden(A, g, g51):
    p, s = demixer51(A).separate_stems(g["wave"])
    assert rel_rms(p, g51["wav_y"].T) < TOL, rel_rms(p, g51["wav_y"].T)
    assert rel_rms(s, g51["wav_v"].T) < TOL, rel_rms(s, g51["wav_v"].T)


def test_v51_tta_oracle(A, g):
    mp = V.small_params_v51()
    sd = V.make_vr51_state(192, 16, 16, 9)
    wp, ws = V.vr_separate_v51(g["wave"][:, :12001], sd, mp, window_size=64, batch_size=2, aggression=10, enable_tta=True, offset=16)
    p, s = demixer51(A, aggression=10, enable_tta=True).separate_stems(g["wave"][:, :12001])
    assert rel_rms(p, wp) < TOL, rel_rms(p, wp)
    assert rel_rms(s, ws) < TOL, rel_rms(s, ws)


def test_high_end_process_golden(A, g):
    p, s = demixer(A, high_end_process=True).separate_stems(g["wave"])
    assert rel_rms(p, g["he_wav_y"].T) < TOL, rel_rms(p, g["he_wav_y"].T)
    assert rel_rms(s, g["he_wav_v"].T) < TOL, rel_rms(s, g["he_wav_v"].T)
