#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ (run in the build container; commit the outputs).

The reference cannot supply expected outputs for this path (cactus_realign is absent from the snapshot and
no reference test holds cigars or posteriors: SURVEY.md 8c, PARITY UNPINNED), so the vectors are produced by
this repository's CPU oracle (oracle/, fp64 log space, plus its fp32 mirror of the device arithmetic) after
it was validated against an independent numpy implementation (tests/test_oracle.py).  Inputs of case
`c1_reference_test_data` are slices of the reference's own test data files
(tests/readFastqFiles/fake_readtype/reads.fq read 1, tests/referenceFastaFiles/reference.fa); only DATA is
kept, no reference source.

    python tests/golden/make_golden.py [/root/reference]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import load_model_arrays, orc, random_pair  # noqa: E402

CODE = {c: i for i, c in enumerate("ACGT")}


def encode(s):
    return np.array([CODE.get(c, 4) for c in s.upper()], dtype=np.uint8)


def needleman_wunsch(X, Y):
    """Plain global alignment (match +1, mismatch -1, gap -1) -> (op,len) guide."""
    n, m = len(X), len(Y)
    S = np.zeros((n + 1, m + 1), dtype=np.int32)
    S[:, 0] = -np.arange(n + 1)
    S[0, :] = -np.arange(m + 1)
    for i in range(1, n + 1):
        sub = np.where(Y == X[i - 1], 1, -1)
        for j in range(1, m + 1):
            S[i, j] = max(S[i - 1, j - 1] + sub[j - 1], S[i - 1, j] - 1, S[i, j - 1] - 1)
    ops = []
    i, j = n, m
    while i > 0 or j > 0:
        if i > 0 and j > 0 and S[i, j] == S[i - 1, j - 1] + (1 if X[i - 1] == Y[j - 1] else -1):
            ops.append(0), (i := i - 1), (j := j - 1)
        elif i > 0 and S[i, j] == S[i - 1, j] - 1:
            ops.append(2), (i := i - 1)
        else:
            ops.append(1), (j := j - 1)
    ops.reverse()
    runs = []
    for o in ops:
        if runs and runs[-1][0] == o:
            runs[-1][1] += 1
        else:
            runs.append([o, 1])
    return [(a, b) for a, b in runs]


def make_case(name, model, X, Y, guide, **kw):
    T, E, _ = load_model_arrays(model)
    h = orc.make_hmm(T, E)
    P = orc.make_params(**kw)
    r64 = orc.realign_read(h, P, X, Y, guide, precision=0)
    r32 = orc.realign_read(h, P, X, Y, guide, precision=1)
    assert r64["status"] == 0 and r32["status"] == 0
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), model=model, X=X, Y=Y, guide=np.array(guide, dtype=np.int32),
        params=np.array([kw.get("band_mode", 0), kw.get("diagonal_expansion", 10), kw.get("constraint_trim", 14),
                         kw.get("split_threshold", 3000), kw.get("fixed_width", 0), kw.get("mode", 0)], dtype=np.int64),
        gammas=np.array([kw.get("gap_gamma", 0.5), kw.get("match_gamma", 0.0)]),
        f64_total_ll=r64["total_ll"], f64_score=r64["score"], f64_ops=np.array(r64["ops"], dtype=np.int32),
        f64_px=r64["px"], f64_py=r64["py"], f64_pp=r64["pp"], cells=r64["cells"],
        f32_total_ll=r32["total_ll"], f32_score=r32["score"], f32_ops=np.array(r32["ops"], dtype=np.int32),
        f32_px=r32["px"], f32_py=r32["py"], f32_pp=r32["pp"].astype(np.float32))
    print("%-28s cells %7d pairs %5d ops %4d ll %.6f  f32==f64 cigar: %s" % (
        name, r64["cells"], len(r64["px"]), len(r64["ops"]), r64["total_ll"], r32["ops"] == r64["ops"]))


def main():
    ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    rng = np.random.default_rng(20260929)
    X, Y, g = random_pair(rng, 150)
    make_case("fixed_w40_150bp", "blasr_hmm_0.txt", X, Y, g, band_mode=1, fixed_width=40)
    X, Y, g = random_pair(rng, 600, indel=0.15, max_indel=6)
    make_case("anchor_default_600bp", "blasr_hmm_0.txt", X, Y, g, band_mode=0, diagonal_expansion=10,
              constraint_trim=14, split_threshold=3000)
    X, Y, g = random_pair(rng, 500, indel=0.2, max_indel=60)
    make_case("anchor_split_500bp", "blasr_hmm_0.txt", X, Y, g, band_mode=0, diagonal_expansion=10,
              constraint_trim=2, split_threshold=15)
    X, Y, g = random_pair(rng, 400)
    make_case("fixed_w200_hmm20_400bp", "blasr_hmm_20.txt", X, Y, g, band_mode=1, fixed_width=200)
    X, Y, g = random_pair(rng, 300)
    make_case("rescore_hmm0_300bp", "blasr_hmm_0.txt", X, Y, g, band_mode=0, split_threshold=100, mode=1)
    X, Y, g = random_pair(rng, 350)
    X[40:44] = 4  # N bases
    make_case("fixed_w100_with_N_350bp", "blasr_hmm_40.txt", X, Y, g, band_mode=1, fixed_width=100,
              gap_gamma=0.3, match_gamma=0.1)
    # C1 plumbing: the reference's own test data (first 700 bases of read 1 vs first 700 of the reference)
    fq = open(os.path.join(ref_root, "tests", "readFastqFiles", "fake_readtype", "reads.fq")).read().split("\n")
    fa = "".join(open(os.path.join(ref_root, "tests", "referenceFastaFiles", "reference.fa")).read().split("\n")[1:])
    Xr, Yr = encode(fa[:700]), encode(fq[1][:700])
    make_case("c1_reference_test_data", "blasr_hmm_0.txt", Xr, Yr, needleman_wunsch(Xr, Yr), band_mode=0,
              diagonal_expansion=10, constraint_trim=14, split_threshold=3000)


if __name__ == "__main__":
    main()
