"""Oracle (oracle/ensemble_oracle.py) pinned on vectors written by the reference Ensembler and spec_utils.invert_stem
(tests/golden/make_golden_ensemble.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ensemble_oracle as E  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "ensemble_small.npz"))


def close(a, b, tol=1e-6):
    assert np.asarray(a).shape == np.asarray(b).shape
    err = np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-30)
    assert err < tol, err


@pytest.mark.parametrize("alg", E.ALGORITHMS)
def test_ensemble_golden(alg):
    w = [G["waves"][k] for k in range(4)]
    close(E.ensemble(w, alg), G[f"{alg}_k4"])
    close(E.ensemble(w[:3], alg), G[f"{alg}_k3"])


def test_weights_and_invert():
    w = [G["waves"][k] for k in range(4)]
    for alg in ("avg_wave", "avg_fft"):
        close(E.ensemble(w, alg, [1.0, 2.0, 0.5, 0.25]), G[f"{alg}_w"])
    close(E.invert_stem(w[0], w[1]), G["invert"])
