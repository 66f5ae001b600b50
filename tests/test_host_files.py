"""CPU tests of the file edges and model-file readers above the C ABI: the WAV codec, the Demucs package / bag reader
(restricted unpickling, checksums, error behaviour of repo.py), the Roformer configuration mirror against the reference's
own normaliser (build container only for that part), CommonSeparator's naming rules."""
import importlib
import importlib.machinery
import logging
import os
import sys
import types

import numpy as np
import pytest

import audio_separator_amd as A
from audio_separator_amd import audio_io, model_files as MF, roformer_config as RC
from tests import separate_cases as SC

REF = "/root/reference"


# ---- WAV codec ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("subtype,tol", [("PCM_16", 2 / 32767), ("PCM_24", 2 / 8388607), ("PCM_32", 1e-6), ("FLOAT", 0.0), ("DOUBLE", 1e-7)])
def test_wav_roundtrip(tmp_path, subtype, tol):
    x = (0.8 * np.random.default_rng(1).standard_normal((1000, 2)).clip(-1, 1)).astype(np.float32)
    p = str(tmp_path / "a.wav")
    audio_io.write_wav(p, x, 48000, subtype)
    info = audio_io.info(p)
    assert (info["samplerate"], info["channels"], info["frames"], info["subtype"]) == (48000, 2, 1000, subtype)
    y, sr = audio_io.read_wav(p)
    assert sr == 48000 and y.shape == (2, 1000) and np.abs(y.T - x).max() <= tol + 1e-9
    assert abs(audio_io.duration(p) - 1000 / 48000) < 1e-9


def test_wav_int16_widening_is_exact(tmp_path):
    pcm = np.array([[-32768, 32767], [1, -1], [12345, -54 + 0]], np.int16)
    for subtype, shift in (("PCM_16", 0), ("PCM_24", 8), ("PCM_32", 16)):
        p = str(tmp_path / f"{subtype}.wav")
        audio_io.write_wav(p, pcm, 44100, subtype)
        y, _ = audio_io.read_wav(p)
        full = float(2 ** (15 + shift))
        assert np.array_equal(np.rint(y.T * full).astype(np.int64), pcm.astype(np.int64) << shift)


def test_load_matches_librosa_contract(tmp_path):
    p = str(tmp_path / "m.wav")
    audio_io.write_wav(p, np.linspace(-0.5, 0.5, 64, dtype=np.float32), 8000, "PCM_16")
    y, sr = audio_io.load(p, sr=8000, mono=False)
    assert y.ndim == 1 and sr == 8000                        # a mono file comes back 1-D, like librosa.load(mono=False)
    with pytest.raises(audio_io.AudioIOError):
        audio_io.load(p, sr=44100)                           # resampling on load is librosa's job
    with pytest.raises(audio_io.AudioIOError):
        open(str(tmp_path / "x.mp3"), "wb").write(b"ID3....")
        audio_io.load(str(tmp_path / "x.mp3"), sr=None)


# ---- CommonSeparator -----------------------------------------------------------------------------------------------------
def _common(**over):
    c = SC.common_config("Some:Model*Name", "/m/some.onnx", {"primary_stem": "Vocals"}, "/tmp/o")
    c.update(over)
    return c


def test_common_separator_names_and_stems():
    s = A.CommonSeparator(_common())
    assert (s.primary_stem_name, s.secondary_stem_name) == ("Vocals", "Instrumental")
    assert s.secondary_stem("No Drums") == "Drums" and s.secondary_stem("Drums") == "No Drums" and s.secondary_stem("") == ""
    assert s.secondary_stem("Primary Stem") == "Secondary Stem" and s.secondary_stem("lead_only") == "backing_only"
    s.audio_file_base = "a<b>c"
    assert s.get_stem_output_path("Vocals", None) == "a_b_c_(Vocals)_Some_Model_Name.wav"
    assert s.get_stem_output_path("Vocals", {"VOCALS": "../x?"}) == "x.wav"
    assert s.sanitize_filename('__a::b//c__. ') == "a_b_c"
    # instruments / target_instrument rules (common_separator.py:104-127)
    t = A.CommonSeparator(_common(model_data={"training": {"instruments": ["other", "vocals"], "target_instrument": "vocals"}}))
    assert (t.primary_stem_name, t.secondary_stem_name) == ("vocals", "other")
    u = A.CommonSeparator(_common(model_data={"training": {"instruments": ["drums"], "target_instrument": None}}))
    assert (u.primary_stem_name, u.secondary_stem_name) == ("drums", "No drums")
    assert not s.is_roformer_model and A.CommonSeparator(_common(model_name="x_Roformer_y")).is_roformer_model
    with pytest.raises(NotImplementedError):
        s.separate("x")
    with pytest.raises(RuntimeError):
        s.write_audio("x.wav", np.zeros((4, 2), np.float32))            # no engine: the writer arithmetic has no CPU path


def test_prepare_mix_contract(tmp_path):
    s = A.CommonSeparator(_common(sample_rate=8000))
    p = str(tmp_path / "z.wav")
    audio_io.write_wav(p, np.zeros((100, 2), np.float32), 8000, "PCM_24")
    with pytest.raises(ValueError, match="empty or not valid"):
        s.prepare_mix(p)
    assert s.input_bit_depth == 24 and s.input_subtype == "PCM_24"
    audio_io.write_wav(p, 0.1 * np.ones(50, np.float32), 8000, "PCM_16")
    m = s.prepare_mix(p)
    assert m.shape == (2, 50) and np.array_equal(m[0], m[1])             # mono -> stereo
    arr = np.random.default_rng(0).standard_normal((30, 2)).astype(np.float32)
    assert np.array_equal(s.prepare_mix(arr), arr.T)


# ---- Demucs packages -------------------------------------------------------------------------------------------------------
def test_demucs_repo_reader(tmp_path):
    repo = SC.write_demucs_repo(str(tmp_path / "repo"))
    bag = MF.get_demucs_model("htd_bag", repo)
    assert bag["is_bag"] and len(bag["models"]) == 2 and bag["segment"] == 1 and bag["weights"][1] == [0.5, 1.5, 1.0, 1.0]
    pkg = bag["models"][0]
    assert pkg["kind"] == "HTDemucs" and pkg["kwargs"]["sources"] == ["drums", "bass", "other", "vocals"] and pkg["kwargs"]["nfft"] == 1024
    assert "demucs" not in sys.modules and "demucs.htdemucs" not in sys.modules        # nothing was imported to read it
    single = MF.get_demucs_model("aaaa1111", repo)
    assert not single["is_bag"] and single["segment"] is None
    with pytest.raises(MF.ModelLoadingError, match="neither a single pre-trained model"):
        MF.get_demucs_model("nope", repo)
    # checksum in the file name is verified (repo.py:check_checksum)
    th = [f for f in os.listdir(repo) if f.startswith("bbbb2222-")][0]
    os.rename(os.path.join(repo, th), os.path.join(repo, "bbbb2222-deadbeef.th"))
    with pytest.raises(MF.ModelLoadingError, match="Invalid checksum"):
        MF.get_demucs_model("htd_bag", repo)
    # models_from_files: segment rules of BagOfModels / demucs_segments
    from audio_separator_amd.demucs import models_from_files
    models, weights = models_from_files(os.path.join(repo, "htd_single.yaml"), "Default")
    assert float(models[0][0].segment) == 1.0 and weights is None
    models, _ = models_from_files(os.path.join(repo, "htd_single.yaml"), "3")
    assert models[0][0].segment == 3 and models[0][0].segment_samples == 24000
    bare = [f for f in os.listdir(repo) if f.startswith("aaaa1111")][0]
    os.rename(os.path.join(repo, bare), os.path.join(repo, "aaaa1111.th"))              # <sig>.th without a checksum suffix
    models, _ = models_from_files(os.path.join(repo, "aaaa1111.th"), "3")
    assert float(models[0][0].segment) == 1.0            # a bare .th is not a bag: demucs_segments leaves it alone


def test_demucs_package_rejects_foreign_globals(tmp_path):
    import torch

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > /dev/null",))
    p = str(tmp_path / "evil.th")
    torch.save({"klass": Evil(), "args": (), "kwargs": {}, "state": {}}, p)
    with pytest.raises(MF.ModelLoadingError):
        MF.read_demucs_package(p)


def test_state_dict_readers_refuse_foreign_globals(tmp_path, monkeypatch):
    """a .pth / .ckpt that needs arbitrary pickle globals is refused by read_state_dict / read_checkpoint -- there is
    no silent retry with weights_only=False; the opt-in is explicit (ASX_ALLOW_UNSAFE_PICKLE=1)."""
    import torch

    marker = tmp_path / "ran"

    class Evil:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))
    p = str(tmp_path / "evil.pth")
    torch.save({"state_dict": {"w": torch.zeros(2)}, "extra": Evil()}, p)
    monkeypatch.delenv(MF.UNSAFE_ENV, raising=False)
    for reader in (MF.read_state_dict, RC.read_checkpoint):
        with pytest.raises(MF.ModelLoadingError, match="refused"):
            reader(p)
    assert not marker.exists()
    good = str(tmp_path / "good.pth")
    torch.save({"state_dict": {"w": torch.ones(3)}}, good)
    assert float(MF.read_state_dict(good)["w"].sum()) == 3.0
    assert float(RC.read_checkpoint(good)["w"].sum()) == 3.0
    monkeypatch.setenv(MF.UNSAFE_ENV, "1")       # the explicit opt-in un-pickles (and therefore runs) the file
    assert "w" in MF.read_state_dict(p)
    assert marker.exists()


def test_damaged_files_are_not_reported_as_refusals(tmp_path, monkeypatch):
    """a missing or truncated download must surface as what it is, not as "refused ... set ASX_ALLOW_UNSAFE_PICKLE=1"."""
    import torch
    monkeypatch.delenv(MF.UNSAFE_ENV, raising=False)
    with pytest.raises(FileNotFoundError):
        MF.safe_torch_load(str(tmp_path / "absent.pth"))
    good = tmp_path / "good.pth"
    torch.save({"w": torch.ones(1000)}, str(good))
    cut = tmp_path / "cut.pth"
    cut.write_bytes(good.read_bytes()[:900])
    with pytest.raises(Exception) as ei:
        MF.safe_torch_load(str(cut))
    assert not isinstance(ei.value, MF.ModelLoadingError) and "UNSAFE" not in str(ei.value)


# ---- Roformer configuration mirror vs the reference's normaliser --------------------------------------------------------------
YAMLS = {
    "ep317": {"audio": {"chunk_size": 352800, "dim_f": 1024, "dim_t": 801, "hop_length": 441, "n_fft": 2048, "num_channels": 2, "sample_rate": 44100},
              "model": {"dim": 512, "depth": 12, "stereo": True, "num_stems": 1, "time_transformer_depth": 1, "freq_transformer_depth": 1,
                        "linear_transformer_depth": 0, "freqs_per_bands": list(RC.DEFAULT_FREQS_PER_BANDS), "dim_head": 64, "heads": 8,
                        "attn_dropout": 0.1, "ff_dropout": 0.1, "flash_attn": True, "dim_freqs_in": 1025, "stft_n_fft": 2048,
                        "stft_hop_length": 441, "stft_win_length": 2048, "stft_normalized": False, "mask_estimator_depth": 3,
                        "multi_stft_resolution_loss_weight": 1.0},
              "training": {"instruments": ["vocals", "other"], "target_instrument": "vocals", "batch_size": 10},
              "inference": {"batch_size": 1, "dim_t": 801, "num_overlap": 4}},
    "mel": {"audio": {"hop_length": 441, "n_fft": 2048, "sample_rate": 44100},
            "model": {"dim": "384", "depth": 6.0, "stereo": "true", "num_stems": 1, "num_bands": 60, "dim_head": 64, "heads": 8,
                      "mask_estimator_depth": 2, "stft_hop_length": 441, "sample_rate": 44100},
            "training": {"instruments": ["vocals", "other"], "target_instrument": "vocals"}, "inference": {"dim_t": 1101}},
    "aliases": {"model": {"dim": 64, "depth": 2, "freq_bands": "(2, 4, 8, 16, 3)", "n_heads": 4, "head_dim": 32, "n_fft": 64, "hop_length": 16,
                          "win_length": 64, "mlp_ratio": 2}, "inference": {"dim_t": 11, "hop_length": 8}},
    "no_hop": {"audio": {"hop_length": 441}, "model": {"dim": 64, "depth": 1, "stereo": True, "freqs_per_bands": [512, 513]}},
}


@pytest.mark.skipif(not os.path.isdir(REF + "/audio_separator"), reason="reference tree not present")
@pytest.mark.parametrize("name", sorted(YAMLS))
def test_roformer_config_matches_reference_normaliser(name, monkeypatch):
    for n, path in (("audio_separator", REF + "/audio_separator"), ("audio_separator.separator", REF + "/audio_separator/separator")):
        pkg = types.ModuleType(n)
        pkg.__path__ = [path]
        pkg.__spec__ = importlib.machinery.ModuleSpec(n, None, is_package=True)
        monkeypatch.setitem(sys.modules, n, pkg)
    norm_mod = importlib.import_module("audio_separator.separator.roformer.configuration_normalizer")
    ref = norm_mod.ConfigurationNormalizer()
    cfg = YAMLS[name]
    for path in ("/m/model_bs_roformer_x.ckpt", "/m/mel_band_roformer_y.ckpt", "/m/plain.ckpt"):
        mine_type = RC.model_type_from_path(cfg, path)
        try:
            want = ref.normalize_from_file_path(cfg, path, apply_defaults=True, validate=True)
        except Exception as e:                                  # reference refuses -> so must the mirror
            with pytest.raises(RC.ParameterValidationError):
                RC.normalize_config(cfg, mine_type)
            assert "ParameterValidationError" in type(e).__name__
            continue
        got = RC.normalize_config(cfg, mine_type)
        assert got == want, (name, path, {k: (got.get(k), want.get(k)) for k in set(got) | set(want) if got.get(k) != want.get(k)})
        assert RC.detect_model_type(got) == ref.detect_model_type(want)
    for k in [k for k in sys.modules if k.startswith("audio_separator.separator.roformer")]:
        sys.modules.pop(k, None)


def test_roformer_constructor_args_quirks():
    c = RC.normalize_config(YAMLS["ep317"], "bs_roformer")
    a = RC.constructor_args(c, "bs_roformer")
    assert a["mask_estimator_depth"] == 2                        # the YAML's 3 never reaches BSRoformer (roformer_loader.py:125-149)
    assert a["stft_hop_length"] == 441 and a["freqs_per_bands"] == RC.DEFAULT_FREQS_PER_BANDS and a["stereo"] is True
    n = RC.constructor_args(RC.normalize_config(YAMLS["no_hop"], "bs_roformer"), "bs_roformer")
    assert n["stft_hop_length"] == 512                           # validator default, not audio.hop_length
    m = RC.constructor_args(RC.normalize_config(YAMLS["mel"], "mel_band_roformer"), "mel_band_roformer")
    assert (m["dim"], m["depth"], m["stereo"], m["mask_estimator_depth"], m["num_bands"]) == (384, 6, True, 2, 60)
    ld = RC.RoformerLoader()
    assert ld.load_model("/nonexistent/bs_roformer.ckpt", {"model": {"depth": 2}}).success is False
    assert ld.validate_configuration({"dim": 8, "depth": 1, "freqs_per_bands": (2, 3)}, "bs_roformer")
    assert not ld.validate_configuration({"dim": 8}, "bs_roformer")
    assert ld.detect_model_type("/x/MelBand_v2.ckpt") == "mel_band_roformer"


def test_roformer_stft_options_follow_the_reference_loader():
    """Where `stft_normalized` / `stft_window_fn` reach the model class in the reference (roformer_loader.py): _create_bs_roformer
    (:123-150) forwards NEITHER; _create_mel_band_roformer (:152-195) forwards both when the configuration has them; the legacy
    fallback (:197-236) passes the whole model section.  The window function becomes the table torch.stft / istft use."""
    import torch
    from audio_separator_amd.mdxc import stft_window_table
    bs = dict(YAMLS["ep317"])
    bs = {**bs, "stft_normalized": True, "stft_window_fn": "torch.hamming_window"}
    a = RC.constructor_args(RC.normalize_config(bs, "bs_roformer"), "bs_roformer")
    assert a["stft_normalized"] is False and a.get("stft_window_fn") is None
    mel = {**YAMLS["mel"], "stft_normalized": True, "stft_window_fn": "torch.hamming_window"}
    m = RC.constructor_args(RC.normalize_config(mel, "mel_band_roformer"), "mel_band_roformer")
    assert m["stft_normalized"] is True and m["stft_window_fn"] == "torch.hamming_window"
    lg = RC.legacy_constructor_args({"dim": 8, "depth": 1, "freqs_per_bands": (2, 3), "stft_normalized": True, "stft_window_fn": torch.blackman_window},
                                    "bs_roformer")
    assert lg["stft_normalized"] is True and lg["stft_window_fn"] is torch.blackman_window
    t = stft_window_table("torch.hamming_window", 48, 64)
    assert t.dtype == np.float32 and t.shape == (64,) and np.all(t[:8] == 0) and np.all(t[56:] == 0)
    assert np.array_equal(t[8:56], torch.hamming_window(48).numpy())
    assert np.array_equal(stft_window_table(torch.blackman_window, 64, 64), torch.blackman_window(64).numpy())
    with pytest.raises(NotImplementedError):
        stft_window_table("os.system", 64, 64)


def test_install_registers_reference_module_names():
    names = A.install()
    try:
        for n in names:
            assert sys.modules[n].__name__.startswith("audio_separator_amd.architectures.")
        assert hasattr(sys.modules["audio_separator.separator.architectures.vr_separator"], "VRSeparator")
    finally:
        A.uninstall()
    assert not any(n in sys.modules for n in names)
