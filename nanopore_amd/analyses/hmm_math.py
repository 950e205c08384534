"""HMM post-processing maths of the reference, restated for Python 3 / numpy.

Mirrors nanopore/analyses/utils.py:611-629 (toMatrix, fromMatrix, normaliseHmmByReferenceGCContent,
modifyHmmEmissionsByExpectedVariationRate, setHmmIndelEmissionsToBeFlat); the same names are re-exported
from nanopore_amd.analyses.utils so call sites read like the reference's.  Known-answer test:
blasr_hmm_20/40.txt == f(blasr_hmm_0.txt) (tests/test_hmm_kat.py).
"""
import numpy as np

from ..hmm import SYMBOL_NUMBER

_K = SYMBOL_NUMBER * SYMBOL_NUMBER


def toMatrix(e):
    """Flat list of 16 -> 4 rows of 4 (row = reference base x, column = read base y); utils.py:611."""
    return [list(e[SYMBOL_NUMBER * i:SYMBOL_NUMBER * (i + 1)]) for i in range(SYMBOL_NUMBER)]


def fromMatrix(m):
    """Inverse of toMatrix; utils.py:612."""
    return [v for row in m for v in row]


def normaliseHmmByReferenceGCContent(hmm, gcContent):
    """Rescale every reference-base row of every state that emits a reference base (all but the insert
    states 2 and 4) so that the reference base frequencies are gc/2 for C,G and (1-gc)/2 for A,T;
    utils.py:614-619."""
    for state in range(hmm.stateNumber):
        if state in (2, 4):
            continue
        block = np.array(hmm.emissions[_K * state:_K * (state + 1)], dtype=np.float64).reshape(SYMBOL_NUMBER, SYMBOL_NUMBER)
        out = np.empty_like(block)
        for x in range(SYMBOL_NUMBER):
            target = gcContent / 2.0 if x in (1, 2) else (1.0 - gcContent) / 2.0
            rowsum = sum(block[x])  # left-to-right float sum, as Python's sum() in the reference
            for y in range(SYMBOL_NUMBER):
                out[x, y] = (block[x, y] / rowsum) * target
        hmm.emissions[_K * state:_K * (state + 1)] = [float(v) for v in out.reshape(-1)]


def modifyHmmEmissionsByExpectedVariationRate(hmm, substitutionRate):
    """match emissions <- match emissions x N, N = (1-r) on the diagonal, r/3 elsewhere; utils.py:621-624."""
    n = np.full((SYMBOL_NUMBER, SYMBOL_NUMBER), substitutionRate / (SYMBOL_NUMBER - 1), dtype=np.float64)
    np.fill_diagonal(n, 1.0 - substitutionRate)
    m = np.array(hmm.emissions[:_K], dtype=np.float64).reshape(SYMBOL_NUMBER, SYMBOL_NUMBER)
    hmm.emissions[:_K] = [float(v) for v in np.dot(m, n).reshape(-1)]


def setHmmIndelEmissionsToBeFlat(hmm):
    """All gap-state emissions <- 1/16; utils.py:626-629."""
    for state in range(1, hmm.stateNumber):
        hmm.emissions[_K * state:_K * (state + 1)] = [1.0 / _K] * _K
