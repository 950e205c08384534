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


def _columns(blocks, ref_len, read_len):
    """Independent restatement of the merge: one letter per alignment column."""
    cols, x, y = [], 0, 0
    for (rp, qp, ops) in blocks:
        cols += "D" * (rp - x) + "I" * (qp - y)
        x, y = rp, qp
        for op, n in ops:
            cols += "MID"[op] * n
            x += n if op != 1 else 0
            y += n if op != 2 else 0
    cols += "D" * (ref_len - x) + "I" * (read_len - y)
    return cols


def test_chain_merge_spells_out_the_gaps_between_blocks():
    """npr_chain_merge (mergeChainedAlignedReads, utils.py:295-386): the global cigar of a chain == the blocks' columns with
    D / I filled in between, before and after; spans == (reference length, read length) (the asserts of :381-382); blocks out
    of order, overlapping or past the sequences are refused as the reference's asserts refuse them."""
    from nanopore_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(8)

    def merge(blocks, ref_len, read_len, cap=None):
        rp = np.array([b[0] for b in blocks], dtype=np.int64)
        qp = np.array([b[1] for b in blocks], dtype=np.int64)
        off = np.concatenate([[0], np.cumsum([len(b[2]) for b in blocks])]).astype(np.int64)
        ops = np.array([o for b in blocks for o in b[2]], dtype=np.int32).reshape(-1, 2)
        out = np.zeros(((int(off[-1]) + 2 * len(blocks) + 2) if cap is None else cap, 2), dtype=np.int32)
        k = L.npr_chain_merge(len(blocks), _lib.ptr(rp), _lib.ptr(qp), _lib.ptr(off), _lib.ptr(ops), ref_len, read_len, _lib.ptr(out), len(out))
        return k, [(int(a), int(b)) for a, b in out[:max(k, 0)]]

    for _ in range(200):
        blocks, x, y = [], 0, 0
        for _ in range(int(rng.integers(1, 6))):
            x += int(rng.integers(0, 30))
            y += int(rng.integers(0, 30))
            ops = [(int(rng.integers(0, 3)), int(rng.integers(1, 20))) for _ in range(int(rng.integers(1, 8)))]
            blocks.append((x, y, ops))
            x += sum(n for op, n in ops if op != 1)
            y += sum(n for op, n in ops if op != 2)
        ref_len, read_len = x + int(rng.integers(0, 25)), y + int(rng.integers(0, 25))
        k, got = merge(blocks, ref_len, read_len)
        assert k == len(got) > 0
        want = _columns(blocks, ref_len, read_len)
        assert "".join("MID"[op] * n for op, n in got) == "".join(want)
        assert all(a[0] != b[0] for a, b in zip(got, got[1:])) and all(n > 0 for _, n in got)   # canonical: merged, no empty ops
        assert sum(n for op, n in got if op != 1) == ref_len and sum(n for op, n in got if op != 2) == read_len
    ok = [(5, 2, [(0, 10)]), (20, 14, [(0, 5)])]
    assert merge(ok, 30, 20)[0] > 0
    assert merge(ok[::-1], 30, 20)[0] == _lib.ERR_INVALID                          # out of chain order
    assert merge([(5, 2, [(0, 10)]), (14, 14, [(0, 5)])], 30, 20)[0] == _lib.ERR_INVALID   # overlap on the reference
    assert merge(ok, 24, 20)[0] == _lib.ERR_INVALID                                # runs past the reference
    assert merge(ok, 30, 20, cap=2)[0] == _lib.ERR_CAPACITY
    assert merge([], 7, 3) == (2, [(2, 7), (1, 3)])                                # no block: everything unaligned
