#!/usr/bin/env python3
"""Modify an HMM output from EM training to normalise for background nucleotide frequencies and to set an
expected substitution rate.  Same command line as the reference's scripts/modifyHmm.py:7-30:

    modifyHmm.py <inputHmm> <gcContent> <substitutionRate> <outputHmm>
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nanopore_amd.analyses.hmm_math import (modifyHmmEmissionsByExpectedVariationRate,  # noqa: E402
                                            normaliseHmmByReferenceGCContent, toMatrix)
from nanopore_amd.hmm import Hmm, SYMBOL_NUMBER  # noqa: E402


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if len(argv) != 5:
        sys.stderr.write(__doc__)
        return 2
    print("ARGS", argv)
    hmm = Hmm.loadHmm(argv[1])
    gcContent = float(argv[2])
    print("Got GC content", gcContent)
    normaliseHmmByReferenceGCContent(hmm, gcContent)
    substitutionRate = float(argv[3])
    print("Got substitution rate", substitutionRate)
    modifyHmmEmissionsByExpectedVariationRate(hmm, substitutionRate)
    k = SYMBOL_NUMBER ** 2
    for state in range(hmm.stateNumber):
        n = np.array(toMatrix(hmm.emissions[k * state:k * (state + 1)]))
        print("For state, ref frequencies", list(n.sum(axis=1)))
        print("For state, read frequencies", list(n.sum(axis=0)))
    hmm.write(argv[4])
    return 0


if __name__ == "__main__":
    sys.exit(main())
