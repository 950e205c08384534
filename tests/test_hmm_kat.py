"""Known-answer tests the reference's own data files provide for the HMM parameter path
(SURVEY.md section 4 / 8c KAT-1): blasr_hmm_20.txt and blasr_hmm_40.txt are
normaliseHmmByReferenceGCContent(0.5) then modifyHmmEmissionsByExpectedVariationRate(0.2 | 0.4) applied to
blasr_hmm_0.txt (nanopore/analyses/utils.py:614-624, scripts/modifyHmm.py:15-22)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import MODEL_DIR, ROOT, load_model_arrays
from nanopore_amd.analyses.hmm_math import (fromMatrix, modifyHmmEmissionsByExpectedVariationRate,
                                            normaliseHmmByReferenceGCContent, setHmmIndelEmissionsToBeFlat,
                                            toMatrix)
from nanopore_amd.hmm import Hmm, SYMBOL_NUMBER, stockHmm


@pytest.mark.parametrize("name", ["blasr_hmm_0.txt", "blasr_hmm_20.txt", "blasr_hmm_40.txt"])
def test_model_file_invariants(name):
    T, E, _ = load_model_arrays(name)
    T = T.reshape(5, 5)
    assert np.allclose(T.sum(axis=1), 1.0, atol=1e-9)
    for s in range(5):
        assert abs(E[16 * s:16 * s + 16].sum() - 1.0) < 1e-9
    # 13 non-zero transitions: match row 5, each gap row 2 (to match, to self)
    assert int((T > 0).sum()) == 13
    # gap states are flat in the shipped files
    assert np.allclose(E[16:], 1.0 / 16.0)


@pytest.mark.parametrize("name", ["blasr_hmm_0.txt", "blasr_hmm_20.txt", "blasr_hmm_40.txt"])
def test_load_write_round_trip_is_textually_identical(name, tmp_path):
    src = os.path.join(MODEL_DIR, name)
    h = Hmm.loadHmm(src)
    assert h.stateNumber == 5 and len(h.emissions) == 80 and len(h.transitions) == 25 and SYMBOL_NUMBER == 4
    out = tmp_path / "hmm.txt"
    h.write(str(out))
    assert out.read_text().split() == open(src).read().split()


@pytest.mark.parametrize("rate,name", [(0.2, "blasr_hmm_20.txt"), (0.4, "blasr_hmm_40.txt")])
def test_kat_modify_hmm(rate, name):
    h = Hmm.loadHmm(os.path.join(MODEL_DIR, "blasr_hmm_0.txt"))
    normaliseHmmByReferenceGCContent(h, 0.5)
    modifyHmmEmissionsByExpectedVariationRate(h, rate)
    T, E, lik = load_model_arrays(name)
    assert np.abs(np.array(h.emissions) - E).max() < 1e-12
    assert h.transitions == list(T)  # transitions untouched, bit-identical
    # match rows (reference base) each sum to 0.25 after GC normalisation at 0.5
    assert np.allclose(np.array(h.emissions[:16]).reshape(4, 4).sum(axis=1), 0.25, atol=1e-12)


def test_modify_hmm_cli_reproduces_shipped_file(tmp_path):
    out = tmp_path / "hmm20.txt"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "modifyHmm.py"),
                           os.path.join(MODEL_DIR, "blasr_hmm_0.txt"), "0.5", "0.2", str(out)],
                          stdout=subprocess.DEVNULL)
    got = np.array(out.read_text().split()[27:], dtype=np.float64)
    _, E, _ = load_model_arrays("blasr_hmm_20.txt")
    assert np.abs(got - E).max() < 2e-12


def test_flat_indels_and_matrix_helpers():
    h = stockHmm()
    setHmmIndelEmissionsToBeFlat(h)
    assert h.emissions[16:] == [1.0 / 16.0] * 64
    m = toMatrix(h.emissions[:16])
    assert len(m) == 4 and all(len(r) == 4 for r in m) and fromMatrix(m) == h.emissions[:16]
    assert m[1][2] == h.emissions[1 * 4 + 2]
    # GC normalisation skips the insert states 2 and 4 (utils.py:617)
    h2 = Hmm.loadHmm(os.path.join(MODEL_DIR, "blasr_hmm_0.txt"))
    h2.emissions[32:48] = [float(i + 1) for i in range(16)]
    before = list(h2.emissions[32:48])
    normaliseHmmByReferenceGCContent(h2, 0.3)
    assert h2.emissions[32:48] == before
    rows = np.array(h2.emissions[16:32]).reshape(4, 4).sum(axis=1)
    assert np.allclose(rows, [0.35, 0.15, 0.15, 0.35])
