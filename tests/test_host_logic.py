"""Host-side stages of libnprealign (band / split planner, MEA chain + cigar, rescore, base encoding) called
through the C ABI WITHOUT a GPU, against the independently written oracle."""
import numpy as np
import pytest

from helpers import cigar_spans, oracle_hmm, orc, random_pair
from nanopore_amd import _lib
from nanopore_amd import realign as R


def _same_plan(a, b):
    return len(a) == len(b) and all(
        all(s[k] == t[k] for k in ("xs", "ys", "xe", "ye", "ragged_start", "ragged_end", "D", "cells"))
        and (s["lo"] == t["lo"]).all() and (s["n"] == t["n"]).all() for s, t in zip(a, b))


@pytest.mark.parametrize("seed", range(4))
def test_planner_equals_oracle(seed):
    rng = np.random.default_rng(400 + seed)
    for it in range(60):
        X, Y, ops = random_pair(rng, int(rng.integers(1, 500)), indel=rng.random() * 0.3, max_indel=int(rng.integers(1, 50)))
        kw = dict(band_mode=it % 2, diagonal_expansion=int(rng.integers(0, 8)) * 2, constraint_trim=int(rng.integers(0, 16)),
                  split_threshold=int(rng.integers(0, 40)), fixed_width=int(rng.integers(2, 300)))
        assert _same_plan(orc.plan(len(X), len(Y), ops, orc.make_params(**kw)),
                          R.plan(R.make_params(**kw), len(X), len(Y), ops)), kw


def test_planner_reference_call_site_parameters():
    """The three parameter sets the reference hard-codes in its call strings (SURVEY.md 8c item vi)."""
    rng = np.random.default_rng(7)
    X, Y, ops = random_pair(rng, 3000, indel=0.1, max_indel=150)
    for split in (3000, 100, 300):  # utils.py:587, alignmentUncertainty.py:41 / marginAlignSnpCaller.py:136, utils.py:511
        kw = dict(band_mode=0, diagonal_expansion=10, constraint_trim=14, split_threshold=split)
        a = orc.plan(len(X), len(Y), ops, orc.make_params(**kw))
        assert _same_plan(a, R.plan(R.make_params(**kw), len(X), len(Y), ops))
    assert len(orc.plan(len(X), len(Y), ops, orc.make_params(split_threshold=100))) >= 1


def test_planner_rejects_non_global_guides():
    P = R.make_params()
    for lX, lY, ops in ((10, 10, [(0, 9)]), (10, 10, [(0, 10), (1, 1)]), (5, 5, [(4, 5)]), (5, 5, [(0, -5)])):
        with pytest.raises(_lib.NprError) as e:
            R.plan(P, lX, lY, ops)
        assert e.value.code == _lib.ERR_INVALID
    # degenerate but legal
    assert R.plan(P, 0, 0, [])[0]["cells"] == 1
    assert R.plan(P, 5, 0, [(2, 5)])[0]["cells"] == 6


@pytest.mark.parametrize("seed", range(5))
def test_mea_and_rescore_equal_oracle(seed):
    rng = np.random.default_rng(500 + seed)
    h = oracle_hmm()
    X, Y, ops = random_pair(rng, int(rng.integers(30, 500)), indel=0.2, max_indel=8)
    seg = orc.plan(len(X), len(Y), ops, orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=80))[0]
    r = orc.fb_f32(h, X, Y, seg["lo"], seg["n"])          # float32 posteriors: what the device hands over
    perm = rng.permutation(len(r["px"]))                   # the C ABI accepts pairs in any order
    for gg, mg in ((0.5, 0.0), (0.0, 0.0), (0.8, 0.2)):
        want, ws = orc.mea_cigar(len(X), len(Y), r["px"], r["py"], r["pp"].astype(np.float64), gg, mg, brute_force=True)
        got, gs = R.mea_cigar(len(X), len(Y), r["px"][perm], r["py"][perm], r["pp"][perm], gg, mg)
        assert got == want and gs == ws
        assert cigar_spans(got) == (len(X), len(Y))
    assert R.rescore(ops, r["px"][perm], r["py"][perm], r["pp"][perm]) == pytest.approx(
        orc.rescore(ops, r["px"], r["py"], r["pp"].astype(np.float64)), abs=1e-15)
    assert R.mea_cigar(4, 3, [], [], [])[0] == [(2, 4), (1, 3)]
    with pytest.raises(_lib.NprError):
        R.mea_cigar(4, 3, [4], [0], [0.5])  # x out of range


def test_encode_bases():
    assert R.encode(b"ACGTacgtNnXU-").tolist() == [0, 1, 2, 3, 0, 1, 2, 3, 4, 4, 4, 4, 4]
