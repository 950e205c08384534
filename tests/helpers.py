"""Shared helpers for the test-suite (test infrastructure; may import oracle/)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402

MODEL_DIR = os.path.join(ROOT, "nanopore_amd", "mappers")


def load_model_arrays(name="blasr_hmm_0.txt"):
    """Independent (test-side) parser of the two-line HMM file: returns (T[25], E[80], likelihood)."""
    with open(os.path.join(MODEL_DIR, name)) as fh:
        l1 = fh.readline().split()
        l2 = fh.readline().split()
    assert len(l1) == 27 and len(l2) == 80
    return np.array(l1[1:26], dtype=np.float64), np.array(l2, dtype=np.float64), float(l1[26])


def oracle_hmm(name="blasr_hmm_0.txt"):
    T, E, _ = load_model_arrays(name)
    return orc.make_hmm(T, E)


def full_matrix_reference(T, E, X, Y, start=None, end=None):
    """Independent O(lX*lY) linear-space, UNBANDED forward/backward in numpy float64.

    Deliberately written in probability space (not log space) and row-by-row (not by anti-diagonal) so
    that it shares no structure with the oracle.  Valid for short sequences only (no scaling).
    Returns (total, posterior[lX, lY]) for the match state.
    """
    T = np.asarray(T).reshape(5, 5)
    Em = np.full((5, 5), 1.0 / 16.0)
    Em[:4, :4] = np.asarray(E[:16]).reshape(4, 4)
    Ex = np.full((5, 5), 0.25)
    Ey = np.full((5, 5), 0.25)
    for s in range(5):
        blk = np.asarray(E[16 * s:16 * s + 16]).reshape(4, 4)
        Ex[s, :4] = blk.sum(axis=1)
        Ey[s, :4] = blk.sum(axis=0)
    lX, lY = len(X), len(Y)
    if start is None:
        start = np.array([1.0, 0, 0, 0, 0])
    if end is None:
        end = T[:, 0].copy()
    F = np.zeros((lX + 1, lY + 1, 5))
    F[0, 0] = start
    for x in range(lX + 1):
        for y in range(lY + 1):
            if x == 0 and y == 0:
                continue
            if x > 0 and y > 0:
                F[x, y, 0] = Em[X[x - 1], Y[y - 1]] * (F[x - 1, y - 1] @ T[:, 0])
            if x > 0:
                for t in (1, 3):
                    F[x, y, t] = Ex[t, X[x - 1]] * (F[x - 1, y] @ T[:, t])
            if y > 0:
                for t in (2, 4):
                    F[x, y, t] = Ey[t, Y[y - 1]] * (F[x, y - 1] @ T[:, t])
    total = F[lX, lY] @ end
    B = np.zeros((lX + 1, lY + 1, 5))
    B[lX, lY] = end
    for x in range(lX, -1, -1):
        for y in range(lY, -1, -1):
            if x == lX and y == lY:
                continue
            acc = np.zeros(5)
            if x < lX and y < lY:
                acc += T[:, 0] * Em[X[x], Y[y]] * B[x + 1, y + 1, 0]
            if x < lX:
                for t in (1, 3):
                    acc += T[:, t] * Ex[t, X[x]] * B[x + 1, y, t]
            if y < lY:
                for t in (2, 4):
                    acc += T[:, t] * Ey[t, Y[y]] * B[x, y + 1, t]
            B[x, y] = acc
    post = F[1:, 1:, 0] * B[1:, 1:, 0] / total
    return total, post, F, B


def random_pair(rng, lX, sub=0.1, indel=0.1, max_indel=3):
    """Random reference X and a noisy copy Y with the TRUE global alignment as (op,len) list."""
    X = rng.integers(0, 4, size=lX).astype(np.uint8)
    Y = []
    ops = []
    x = 0
    while x < lX:
        r = rng.random()
        if r < indel / 2:
            k = int(rng.integers(1, max_indel + 1))
            k = min(k, lX - x)
            ops.append((2, k))
            x += k
        elif r < indel:
            k = int(rng.integers(1, max_indel + 1))
            Y.extend(rng.integers(0, 4, size=k).tolist())
            ops.append((1, k))
        else:
            b = int(X[x])
            if rng.random() < sub:
                b = (b + int(rng.integers(1, 4))) % 4
            Y.append(b)
            ops.append((0, 1))
            x += 1
    merged = []
    for op, k in ops:
        if merged and merged[-1][0] == op:
            merged[-1] = (op, merged[-1][1] + k)
        else:
            merged.append((op, k))
    return X, np.array(Y, dtype=np.uint8), merged


def cigar_spans(ops):
    sx = sum(k for op, k in ops if op in (0, 2))
    sy = sum(k for op, k in ops if op in (0, 1))
    return sx, sy


def seg_arith_of(batch):
    """i -> which fp32 arithmetic the device ran each segment of read i in (Batch.segment_arith: 0 per-cell exponents, 1
    row-scaled), for orc.realign_read(..., precision=1, seg_arith=...): the mirror restates whichever the kernel class used."""
    off, ar = batch.segment_arith()
    return lambda i: ar[off[i]:off[i + 1]]
