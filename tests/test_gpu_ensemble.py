"""GPU parity of the spectral edges against golden vectors written by the reference Ensembler / spec_utils.invert_stem."""
import os

import numpy as np
import pytest

from oracle import ensemble_oracle as E

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import audio_separator_amd as A
    return A.Engine(A.MDXConfig(n_fft=64, hop_length=16, dim_f=32, segment_size=8))


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "ensemble_small.npz"))


def rel(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("alg", E.ALGORITHMS)
def test_ensemble_golden(eng, g, alg):
    w = [g["waves"][k] for k in range(4)]
    tol = 0.0 if alg in ("median_wave", "min_wave", "max_wave", "ensemble_wav") else 5e-6
    assert rel(eng.ensemble(w, alg), g[f"{alg}_k4"]) <= tol
    assert rel(eng.ensemble(w[:3], alg), g[f"{alg}_k3"]) <= tol


def test_weights_ragged_and_invert(eng, g):
    w = [g["waves"][k] for k in range(4)]
    for alg in ("avg_wave", "avg_fft"):
        assert rel(eng.ensemble(w, alg, [1.0, 2.0, 0.5, 0.25]), g[f"{alg}_w"]) < 5e-6
    assert rel(eng.invert_stem(w[0], w[1]), g["invert"]) < 1e-5
    # shorter inputs are zero padded to the longest (ensembler.py:29-30)
    ragged = [w[0], w[1][:, :7000], w[2][:, :8123]]
    want = E.ensemble([np.pad(x, ((0, 0), (0, 9001 - x.shape[1]))) for x in ragged], "max_fft")
    assert rel(eng.ensemble(ragged, "max_fft"), want) < 5e-6
    with pytest.raises(ValueError):
        eng.ensemble(w, "nope")
