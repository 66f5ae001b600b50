"""Drop-in parity on the MI355X: this repo's MDXSeparator / MDXCSeparator / DemucsSeparator / VRSeparator, constructed like
``Separator.load_model`` constructs the reference's (``cls(common_config=..., arch_config=...)``) and called like
``_separate_file`` calls them (``separate(path, custom_output_names)``), against goldens written by the REFERENCE's own
``separate()`` on the same files and model files (tests/golden/make_golden_separate.py).

Compared: the returned file names, the order and names of the writes, every array handed to ``write_audio``
(relative RMS <= 1e-4, the north-star stem tolerance, fp32), and the PCM16 WAV files actually written (within 2 LSB + 3e-4
of the reference's array pushed through the reference's writer arithmetic).  Everything numerical goes through libasx.so.
"""
import json
import os

import numpy as np
import pytest

from tests import separate_cases as SC

pytestmark = pytest.mark.gpu


def _run(family, tmp_path, monkeypatch, tol=SC.TOL_STEM):
    g = np.load(os.path.join(SC.GOLDEN, f"separate_{family}.npz"))
    out = []
    for case in SC.cases(family, str(tmp_path)):
        inst, worst = SC.run_case(case, g, monkeypatch, tol=tol)
        assert inst.engine is not None and type(inst.engine).__name__ == "Engine"
        out.append((case[0], inst, worst))
    return g, out


def test_mdx_separator_dropin(tmp_path, monkeypatch):
    _, res = _run("mdx", tmp_path, monkeypatch)
    inst = res[0][1]
    assert (inst.n_bins, inst.trim, inst.chunk_size, inst.gen_size) == (49, 48, 240, 144)


def test_mdxc_separator_dropin(tmp_path, monkeypatch):
    g, res = _run("mdxc", tmp_path, monkeypatch)
    for tag, inst, _ in res:
        assert bool(inst.override_model_segment_size) == bool(g[f"{tag}__override"])


def test_roformer_separator_dropin(tmp_path, monkeypatch):
    g, res = _run("roformer", tmp_path, monkeypatch)
    assert res[0][1].get_roformer_loading_stats() == json.loads(str(g["rof__stats"]))


def test_demucs_separator_dropin(tmp_path, monkeypatch):
    _run("demucs", tmp_path, monkeypatch)


def test_vr_separator_dropin(tmp_path, monkeypatch):
    _run("vr", tmp_path, monkeypatch)


def test_mdx_invert_using_spec(tmp_path, monkeypatch):
    """invert_using_spec=True: secondary = spec_utils.invert_stem(demix(mix, match), primary * compensate) with the stem as
    [2, N] (the reference's own call passes [N, 2] and cannot run, mdx_separator.py:177-179); checked against the oracle's
    restatement of invert_stem on this engine's own primary."""
    from oracle import ensemble_oracle as EO
    case = SC.cases("mdx", str(tmp_path))[0]
    tag, cls, common, arch, wav, custom = case
    common = dict(common, invert_using_spec=True)
    inst = SC.plugin_class(cls)(common_config=common, arch_config=arch)
    calls = []
    inst.write_audio = lambda p, a: calls.append((p, np.array(a, copy=True)))
    names = inst.separate(wav, None)
    assert len(names) == 2 and len(calls) == 2
    secondary, primary = calls[0][1], calls[1][1]
    from audio_separator_amd import audio_io
    mix, _ = audio_io.read_wav(wav)
    peak = np.abs(mix).max()
    mix = mix * np.float32(0.9 / peak) if peak > 0.9 else mix
    raw = inst.demix(np.ascontiguousarray(mix, np.float32), is_match_mix=True)
    want = EO.invert_stem(raw, (primary * inst.compensate).T)
    assert secondary.shape == want.shape
    assert SC.rel_rms(secondary, want) < 1e-4


def test_normalize_bit_exact_and_inplace():
    """asx_normalize = spec_utils.normalize (uvr_lib_v5/spec_utils.py:99-115): float32 numpy semantics, in place."""
    import audio_separator_amd as A
    eng = A.Engine(A.MDXConfig(n_fft=96, hop_length=16, dim_f=32, segment_size=16))
    rng = np.random.default_rng(3)
    for amp, thr, amp_thr in ((2.0, 0.9, None), (0.2, 0.9, 0.5), (0.5, 0.9, 0.0), (0.95, 1.0, None)):
        x = (amp * rng.uniform(-1, 1, (2, 5001))).astype(np.float32)
        want = x.copy()
        maxv = np.abs(want).max()
        if maxv > thr:
            want *= thr / maxv
        elif amp_thr is not None and maxv < amp_thr:
            want *= amp_thr / maxv
        got = eng.normalize(x, thr, amp_thr)
        assert got is x and np.array_equal(x, want)
    v = np.asfortranarray((3.0 * rng.uniform(-1, 1, (700, 2))).astype(np.float32)).T      # non-contiguous view: written back
    ref = v.copy()
    ref *= np.float32(0.9) / np.abs(ref).max()
    out = eng.normalize(v, 0.9, 0.0)
    assert np.array_equal(np.asarray(out), ref)


def test_soundfile_writer_path_keeps_input_subtype(tmp_path, monkeypatch):
    """use_soundfile=True (common_separator.py:399-461): normalise on the device, keep the input's subtype (24-bit in -> PCM_24 out)."""
    from audio_separator_amd import audio_io
    case = SC.cases("mdx", str(tmp_path))[0]
    tag, cls, common, arch, wav, custom = case
    x, sr = audio_io.read_wav(wav)
    wav24 = str(tmp_path / "in24.wav")
    audio_io.write_wav(wav24, x.T, sr, "PCM_24")
    inst = SC.plugin_class(cls)(common_config=dict(common, use_soundfile=True), arch_config=arch)
    names = inst.separate(wav24, None)
    assert inst.input_bit_depth == 24 and inst.input_subtype == "PCM_24"
    for n in names:
        info = audio_io.info(os.path.join(common["output_dir"], n))
        assert info["subtype"] == "PCM_24" and info["channels"] == 2 and info["frames"] == x.shape[1]
        y, _ = audio_io.read_wav(os.path.join(common["output_dir"], n))
        assert 0.1 < np.abs(y).max() <= 0.9 + 1e-6


def _separate_once(case, wav, **common_over):
    tag, cls, common, arch, _, custom = case
    inst = SC.plugin_class(cls)(common_config=dict(common, asx_profile_file=True, **common_over), arch_config=arch)
    names = inst.separate(wav, None)
    srcs = (np.array(inst.secondary_source, copy=True), np.array(inst.primary_source, copy=True))
    timings = dict(inst.file_timings)
    blobs = []
    for n in names:
        with open(os.path.join(common["output_dir"], n), "rb") as f:
            blobs.append(f.read())
    inst.clear_gpu_cache()
    inst.clear_file_specific_paths()
    assert inst._dev_stems == {}
    return names, srcs, blobs, timings


@pytest.mark.parametrize("subtype,channels", [("PCM_16", 2), ("PCM_24", 2), ("PCM_32", 2), ("FLOAT", 2), ("PCM_16", 1)])
def test_device_resident_file_path_equals_host_path(tmp_path, monkeypatch, subtype, channels):
    """The file-level fast path (data chunk -> pinned -> HBM -> asx_pcm_decode_dev -> asx_separate_dev -> asx_pcm16_rows_dev,
    stems never re-uploaded) against the generic path (host decode, host stems, asx_pcm16 per stem): the same stem arrays and
    the same stem FILES, byte for byte, for every WAVE sample format and for a mono input."""
    from audio_separator_amd import audio_io
    case = SC.cases("mdx", str(tmp_path))[0]
    x, sr = audio_io.read_wav(case[4])
    if channels == 1:
        x = x[:1]
    src = str(tmp_path / f"in_{subtype}_{channels}.wav")
    audio_io.write_wav(src, np.ascontiguousarray(x.T), sr, subtype)
    monkeypatch.setenv("ASX_FILE_FASTPATH", "1")
    names_d, srcs_d, blobs_d, t_d = _separate_once(case, src)
    assert "h2d_decode" in t_d and "demix" in t_d, t_d            # the device path ran
    monkeypatch.setenv("ASX_FILE_FASTPATH", "0")
    names_h, srcs_h, blobs_h, t_h = _separate_once(case, src)
    assert "h2d_decode" not in t_h
    assert names_d == names_h
    for a, b in zip(srcs_d, srcs_h):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert blobs_d == blobs_h


def test_device_path_silent_file_raises_like_prepare_mix(tmp_path):
    from audio_separator_amd import audio_io
    case = SC.cases("mdx", str(tmp_path))[0]
    tag, cls, common, arch, _, _ = case
    src = str(tmp_path / "silent.wav")
    audio_io.write_wav(src, np.zeros((4000, 2), np.int16), 44100, "PCM_16")
    inst = SC.plugin_class(cls)(common_config=common, arch_config=arch)
    with pytest.raises(ValueError, match="empty or not valid"):
        inst.separate(src, None)


def test_foreign_array_still_takes_the_upload_path(tmp_path):
    """write_audio with an array separate() did not produce (the ensembler's output, separator.py:1379) must not hit the
    device cache: same file as eng.pcm16 of that array."""
    from audio_separator_amd import audio_io
    case = SC.cases("mdx", str(tmp_path))[0]
    tag, cls, common, arch, wav, _ = case
    inst = SC.plugin_class(cls)(common_config=common, arch_config=arch)
    inst.separate(wav, None)
    foreign = np.array(inst.primary_source, copy=True) * np.float32(0.5)
    #  a written stem's device tensor and pinned mirror are released with its registry entry, and the mirror that
    # stays in ``primary_source`` is read-only: an in-place edit raises instead of being silently ignored by a later write_audio
    assert inst._device_stem_for(foreign) is None and inst._device_stem_for(inst.primary_source) is None
    assert not inst.primary_source.flags.writeable
    with pytest.raises(ValueError):
        inst.primary_source[0, 0] = 1.0
    inst.write_audio("foreign.wav", foreign)
    pcm, _ = audio_io.read_wav(os.path.join(common["output_dir"], "foreign.wav"))
    want, _ = inst.engine.pcm16(foreign, 0.9, 0.0)
    assert np.array_equal(np.rint(pcm.T * 32768.0).astype(np.int16), want)


@pytest.mark.parametrize("idx", [0, 1])
def test_demucs_device_resident_file_path_equals_host_path(tmp_path, monkeypatch, idx):
    """DemucsSeparator (single model and a two-member bag): stems that stay in HBM between the demix and the writer's int16 pass
    (asx_pcm16_dev on the planar device stem) against the generic path (host decode, host stems, upload per stem): same stem
    files byte for byte, same draws of the random shifts on both sides."""
    import random
    case = SC.cases("demucs", str(tmp_path))[idx]
    tag, cls, common, arch, wav, custom = case
    seq = [1234, 777, 3999, 42, 2500, 9]

    def run(fast):
        it = iter(seq)
        monkeypatch.setattr(random, "randint", lambda a, b: next(it))
        monkeypatch.setenv("ASX_FILE_FASTPATH", "1" if fast else "0")
        inst = SC.plugin_class(cls)(common_config=dict(common, asx_profile_file=True), arch_config=arch)
        names = inst.separate(wav, None)
        t = dict(inst.file_timings)
        blobs = []
        for n in names:
            with open(os.path.join(common["output_dir"], n), "rb") as f:
                blobs.append(f.read())
        inst.clear_gpu_cache()
        inst.clear_file_specific_paths()
        return names, blobs, t
    nd, bd, td = run(True)
    assert "h2d_decode" in td and "demix" in td, td
    nh, bh, th = run(False)
    assert "h2d_decode" not in th
    assert nd == nh and bd == bh


def test_vr_device_resident_file_path_equals_host_path(tmp_path, monkeypatch):
    """VRSeparator: the wave decoded on the device, both stems kept in HBM until the int16 pass -- same files as the generic path."""
    case = SC.cases("vr", str(tmp_path))[0]
    tag, cls, common, arch, wav, custom = case

    def run(fast):
        monkeypatch.setenv("ASX_FILE_FASTPATH", "1" if fast else "0")
        inst = SC.plugin_class(cls)(common_config=dict(common, asx_profile_file=True), arch_config=arch)
        names = inst.separate(wav, None)
        t = dict(inst.file_timings)
        blobs = []
        for n in names:
            with open(os.path.join(common["output_dir"], n), "rb") as f:
                blobs.append(f.read())
        inst.clear_gpu_cache()
        inst.clear_file_specific_paths()
        return names, blobs, t, (inst.input_subtype, inst.input_bit_depth)
    nd, bd, td, sd = run(True)
    assert "h2d_decode" in td and "demix" in td, td
    nh, bh, th, sh = run(False)
    assert "h2d_decode" not in th
    assert nd == nh and bd == bh and sd == sh


@pytest.mark.parametrize("family,idx", [("mdxc", 0), ("mdxc", 1), ("mdxc", 2), ("roformer", 0)])
def test_mdxc_device_resident_file_path_equals_host_path(tmp_path, monkeypatch, family, idx):
    """MDXCSeparator (TFC-TDF v3 with two targets / one target + residual / single stem, and a Roformer): decode, normalise, demix,
    residual and per-stem normalise in HBM (asx_pcm_decode_dev, asx_normalize_dev, asx_residual_dev) against the generic path:
    byte-identical stem files."""
    case = SC.cases(family, str(tmp_path))[idx]
    tag, cls, common, arch, wav, custom = case

    def run(fast):
        monkeypatch.setenv("ASX_FILE_FASTPATH", "1" if fast else "0")
        inst = SC.plugin_class(cls)(common_config=dict(common, asx_profile_file=True), arch_config=arch)
        names = inst.separate(wav, custom)
        t = dict(inst.file_timings)
        blobs = []
        for n in names:
            with open(os.path.join(common["output_dir"], n), "rb") as f:
                blobs.append(f.read())
        inst.clear_gpu_cache()
        inst.clear_file_specific_paths()
        return names, blobs, t
    nd, bd, td = run(True)
    assert "h2d_decode" in td and "demix" in td, td
    nh, bh, th = run(False)
    assert "h2d_decode" not in th
    assert nd == nh and bd == bh
