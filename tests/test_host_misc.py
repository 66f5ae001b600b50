"""CPU tests of small host-side helpers of the mirrors (no engine): the VR model-parameter / capacity lookups and the Mel-Band
Roformer band layout the product computes on its own (it must agree with the oracle's restatement bin for bin)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from audio_separator_amd import vr as VR  # noqa: E402
from audio_separator_amd.mdxc import mel_band_layout  # noqa: E402
from oracle import roformer_oracle as R  # noqa: E402


@pytest.mark.parametrize("sr,n_fft,n_mels", [(44100, 2048, 60), (44100, 2048, 64), (8000, 256, 12), (100, 64, 5)])
def test_mel_band_layout_matches_oracle(sr, n_fft, n_mels):
    ps, pc = mel_band_layout(sr, n_fft, n_mels)
    os_, oc = R.mel_band_layout(sr, n_fft, n_mels)
    assert (list(ps), list(pc)) == (list(os_), list(oc))
    assert ps[0] == 0 and ps[-1] + pc[-1] == n_fft // 2 + 1          # bin 0 and the last bin are forced in
    assert all(ps[i] <= ps[i + 1] <= ps[i] + pc[i] for i in range(n_mels - 1))   # bands overlap or touch, never leave a gap


def test_load_model_params_like_the_reference(tmp_path):
    # ModelParameters (model_param_init.py:48-71): digit keys become ints at every level, missing flags default to False,
    # n_bins is an alias of bins
    p = {"n_bins": 672, "unstable_bins": 8, "reduction_bins": 530, "band": {"1": {"sr": 7350, "hl": 80, "n_fft": 640}, "2": {"sr": 44100}},
         "sr": 44100, "pre_filter_start": 668, "pre_filter_stop": 672, "mid_side": True}
    f = tmp_path / "4band.json"
    f.write_text(json.dumps(p))
    got = VR.load_model_params(str(f))
    assert set(got["band"].keys()) == {1, 2} and got["band"][1]["hl"] == 80
    assert got["bins"] == 672 and got["mid_side"] is True and got["reverse"] is False and got["mid_side_b2"] is False
    same = VR.load_model_params({"bins": 5, "band": {1: {}}})
    assert same["bins"] == 5 and same["stereo_w"] is False


def test_arch_size_and_capacity(tmp_path):
    # vr_separator.py:161-164 picks the known size nearest to the file size in KiB; nets.py:65-93 maps it to the layer widths
    f = tmp_path / "model.pth"
    f.write_bytes(b"\0" * (123000 * 1024))
    assert VR.nn_arch_size_from_file(str(f)) == 123812
    assert VR.model_capacity(123821)[0] == (2, 32) and VR.model_capacity(537227)[5] == (64, 128) and VR.model_capacity(31191)[2] == (18, 8, 1, 1, 0)
    with pytest.raises(NotImplementedError):
        VR.model_capacity(56817)      # a VR 5.1 size: no capacity table, nout / nout_lstm come from model_data
