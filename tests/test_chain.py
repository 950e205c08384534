"""Native chaining (include/nprealign.h: npr_chain_hits; nanopore_amd/csrc/npr_chain.cpp) against a test-side restatement
of the reference's all-pairs scan (nanopore/analyses/utils.py:388-426) on random hit sets, ties included."""
import numpy as np
import pytest

from nanopore_amd import _lib


def _quadratic(rs, qs, re, qe, rev, score, max_gap):
    n = len(rs)
    order = sorted(range(n), key=lambda i: rs[i])
    best = list(score)
    back = {}
    for a, i in enumerate(order):
        for j in order[:a]:
            if rs[i] > re[j] and qs[i] > qe[j] and rev[i] == rev[j] and rs[i] - re[j] + qs[i] - qe[j] <= max_gap and score[i] + best[j] > best[i]:
                best[i] = score[i] + best[j]
                back[i] = j
    i = sorted(order, key=lambda k: best[k])[-1]
    chain = [i]
    while i in back:
        i = back[i]
        chain.append(i)
    return chain[::-1]


def _native(rs, qs, re, qe, rev, score, max_gap):
    L = _lib.load()
    n = len(rs)
    a = [np.ascontiguousarray(v, dtype=np.int64) for v in (rs, qs, re, qe)]
    r = np.ascontiguousarray(rev, dtype=np.uint8)
    s = np.ascontiguousarray(score, dtype=np.int64)
    out = np.zeros(max(n, 1), dtype=np.int64)
    k = L.npr_chain_hits(n, _lib.ptr(a[0]), _lib.ptr(a[1]), _lib.ptr(a[2]), _lib.ptr(a[3]), _lib.ptr(r), _lib.ptr(s), max_gap, _lib.ptr(out))
    assert k >= 0
    return [int(v) for v in out[:k]]


@pytest.mark.parametrize("seed", range(6))
def test_native_chain_equals_the_all_pairs_scan(seed):
    rng = np.random.default_rng(3000 + seed)
    for _ in range(60):
        n = int(rng.integers(1, 120))
        span = int(rng.choice([300, 2000, 20000]))
        rs = rng.integers(0, span, size=n)
        ln = rng.integers(1, 60, size=n)
        shift = rng.integers(-30, 30, size=n) if seed % 2 else np.zeros(n, dtype=np.int64)
        rev = (rng.random(n) < 0.3).astype(int)
        qs = rs + shift + rng.integers(-3, 4, size=n)
        qs = np.where(rev == 1, -qs - ln, qs)   # signed read positions of reverse-strand hits are negative
        re, qe = rs + ln - 1, qs + ln - 1
        score = rng.integers(1, 4, size=n) if seed == 0 else ln    # small scores force ties
        for max_gap in (200, 20):
            assert _native(rs, qs, re, qe, rev, score, max_gap) == _quadratic(list(rs), list(qs), list(re), list(qe), list(rev), [int(v) for v in score], max_gap)


def test_degenerate_inputs():
    assert _native([], [], [], [], [], [], 200) == []
    assert _native([5], [7], [9], [11], [0], [5], 200) == [0]
    # equal chains: the end that sorts last wins, the predecessor that sorts first wins
    assert _native([0, 0, 10], [0, 0, 10], [4, 4, 14], [4, 4, 14], [0, 0, 0], [5, 5, 5], 200) == [0, 2]
