"""Host logic of the wide-band kernel (k_dp_tile, nanopore_amd/csrc/npr_kernel_tile.hip), checked without a GPU.

The kernel cuts the lattice columns of a segment into stripes and sweeps each stripe anti-diagonal by anti-diagonal with
slot j = column X + j.  This file replays that schedule step by step on the CPU -- the same slot geometry, the same
neighbour-cell hand-over between stripes, the same read-base stream indices, the right-aligned backward stripes with
their lane offset into the forward rows -- with an integer stand-in for the cell arithmetic, and compares every cell and
every emitted pair with a plain evaluation of the recurrences over the band.  (The arithmetic itself is compared on the
GPU, tests/test_gpu_tile.py; a cell's value does not depend on the order of evaluation.)"""
import numpy as np
import pytest

from helpers import random_pair
from nanopore_amd import realign as R

P = 1000003


def _fwd(L, M, U, bx, by):
    return (3 * L + 5 * M + 7 * U + 11 * bx + 13 * by + 1) % P


def _bwd(Ms, Xs, Ys, bx, by):
    return (2 * Ms + 9 * Xs + 4 * Ys + 17 * bx + 19 * by + 1) % P


def _base(seq, i):
    return int(seq[i]) if 0 <= i < len(seq) else 4


def _direct(X, Y, lo, n):
    """Plain evaluation: forward / backward values of every band cell, dead (0 contribution) outside."""
    lX, lY = len(X), len(Y)
    inb = {}
    for d in range(lX + lY + 1):
        for j in range(n[d]):
            xmy = lo[d] + 2 * j
            inb[((d + xmy) // 2, (d - xmy) // 2)] = True
    F, B = {}, {}
    for d in range(lX + lY + 1):
        for (x, y) in [((d + lo[d] + 2 * j) // 2, (d - lo[d] - 2 * j) // 2) for j in range(n[d])]:
            if d == 0:
                F[(x, y)] = 12345
            else:
                F[(x, y)] = _fwd(F.get((x - 1, y), 0), F.get((x - 1, y - 1), 0), F.get((x, y - 1), 0), _base(X, x - 1), _base(Y, y - 1))
    D = lX + lY
    for d in range(D, -1, -1):
        for (x, y) in [((d + lo[d] + 2 * j) // 2, (d - lo[d] - 2 * j) // 2) for j in range(n[d])]:
            if d == D:
                B[(x, y)] = 54321
            else:
                B[(x, y)] = _bwd(B.get((x + 1, y + 1), 0), B.get((x + 1, y), 0), B.get((x, y + 1), 0), _base(X, x), _base(Y, y))
    pairs = {(x - 1, y - 1): (F[(x, y)] * B[(x, y)]) % P for (x, y) in F if x + y >= 2}
    return F, B, pairs


def _tile(X, Y, lo, n, st, Rr):
    """The kernel's schedule: stripes in column order (forward), then in reverse (backward)."""
    K = 64 * Rr
    lX, lY = len(X), len(Y)
    D = lX + lY
    S = len(st["X"])
    rows = {}   # forward rows: rows[row][slot] (left-aligned layout)
    edge_f = {}  # row -> last column's cell
    F, B, pairs = {}, {}, {}

    def band(d, origin, c0, c1):
        xlo = (d + lo[d]) >> 1
        j0, j1 = max(xlo - origin, c0), min(xlo + n[d] - 1 - origin, c1)
        return (j0, j1 - j0 + 1) if j1 >= j0 else (0, 0)

    for s in range(S):
        Xs, Ks, df, dl, row0 = (int(st[k][s]) for k in ("X", "K", "df", "dl", "row0"))
        if dl < df:
            continue
        dfL, dlL, row0L = (int(st["df"][s - 1]), int(st["dl"][s - 1]), int(st["row0"][s - 1])) if s else (1, 0, 0)
        bx = [_base(X, Xs + j - 1) for j in range(K)]
        by = [_base(Y, (df - 1) - Xs - j - 1) for j in range(K)]
        p1, p2 = [0] * K, [0] * K          # anti-diagonals d-1, d-2 (unshifted)
        q0 = df - 2 - dfL                  # (x-1, y-1) of slot 0 on the first anti-diagonal: the left stripe's cell on df-2
        carry = edge_f[row0L + q0] if 0 <= q0 <= dlL - dfL else 0
        for d in range(df, dl + 1):
            jlo, nn = band(d, Xs, 0, Ks - 1)
            q = d - 1 - dfL
            edge = edge_f[row0L + q] if 0 <= q <= dlL - dfL else 0
            by = [_base(Y, d - Xs - 1)] + by[:-1]                      # bases_down, inject at slot 0
            new = [0] * K
            for j in range(K):
                Lc = p1[j - 1] if j else edge
                Mc = p2[j - 1] if j else carry
                v = _fwd(Lc, Mc, p1[j], bx[j], by[j])
                new[j] = v if jlo <= j < jlo + nn else 0
            carry = edge
            if d == 0:
                new[0] = 12345
            row = row0 + d - df
            rows[row] = list(new)
            edge_f[row] = new[Ks - 1]
            for j in range(jlo, jlo + nn):
                F[(Xs + j, d - Xs - j)] = new[j]
            p2, p1 = p1, new
    edge_b = {}
    for s in range(S - 1, -1, -1):
        Xs, Ks, df, dl, row0 = (int(st[k][s]) for k in ("X", "K", "df", "dl", "row0"))
        if dl < df:
            continue
        dfR, dlR, row0R = (int(st["df"][s + 1]), int(st["dl"][s + 1]), int(st["row0"][s + 1])) if s + 1 < S else (1, 0, 0)
        pad = K - Ks
        X0 = Xs - pad
        bx = [_base(X, X0 + j) for j in range(K)]
        by = [_base(Y, (dl + 1) - X0 - j) for j in range(K)]
        s1, s2 = [0] * K, [0] * K
        q0 = dl + 2 - dfR                  # (x+1, y+1) of the top slot on the first anti-diagonal: the right stripe's cell on dl+2
        carry = edge_b[row0R + q0] if 0 <= q0 <= dlR - dfR else 0
        for d in range(dl, df - 1, -1):
            jlo, nn = band(d, X0, pad, K - 1)
            q = d + 1 - dfR
            edge = edge_b[row0R + q] if 0 <= q <= dlR - dfR else 0
            by = by[1:] + [_base(Y, d - X0 - (K - 1))]                 # bases_up, inject at the top slot
            new = [0] * K
            for j in range(K):
                Xc = s1[j + 1] if j + 1 < K else edge
                Mc = s2[j + 1] if j + 1 < K else carry
                v = _bwd(Mc, Xc, s1[j], bx[j], by[j])
                new[j] = v if jlo <= j < jlo + nn else 0
            carry = edge
            if d == D:
                new[lX - X0] = 54321
            row = row0 + d - df
            edge_b[row] = new[pad]
            frow = rows[row]
            for j in range(jlo, jlo + nn):
                x, y = X0 + j, d - X0 - j
                B[(x, y)] = new[j]
                if d >= 2:
                    pairs[(x - 1, y - 1)] = (frow[j - pad] * new[j]) % P   # forward row read with the lane offset
            s2, s1 = s1, new
    return F, B, pairs


CASES = [
    dict(n=900, params=dict(band_mode=R.BAND_FIXED, fixed_width=420), indel=0.2, max_indel=30),
    dict(n=700, params=dict(band_mode=R.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=3000), indel=0.25, max_indel=12),
    dict(n=260, params=dict(band_mode=R.BAND_FIXED, fixed_width=40), indel=0.1, max_indel=5),
    dict(n=130, params=dict(band_mode=R.BAND_ANCHOR, diagonal_expansion=6, constraint_trim=3, split_threshold=3000), indel=0.3, max_indel=60),
    dict(n=5, params=dict(band_mode=R.BAND_FIXED, fixed_width=10), indel=0.0, max_indel=1),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("Rr", [2, 4])
def test_stripe_schedule_reproduces_the_recurrences(case, Rr):
    c = CASES[case]
    rng = np.random.default_rng(100 + case)
    X, Y, g = random_pair(rng, c["n"], indel=c["indel"], max_indel=c["max_indel"])
    params = R.make_params(**c["params"])
    segs = R.plan(params, len(X), len(Y), g)
    assert len(segs) == 1
    lo, n = segs[0]["lo"], segs[0]["n"]
    st = R.stripes(params, len(X), len(Y), g, slots_per_lane=Rr)
    K = 64 * Rr
    assert len(st["X"]) == len(X) // K + 1 and st["df"][0] == 0 and st["dl"][-1] == len(X) + len(Y)
    assert st["rows"] == int(np.maximum(st["dl"] - st["df"] + 1, 0).sum())
    F0, B0, P0 = _direct(X, Y, lo, n)
    F1, B1, P1 = _tile(X, Y, lo, n, st, Rr)
    assert F0 == F1
    assert B0 == B1
    assert P0 == P1
