"""The CPU oracle checked against an independent implementation and against its own invariants.

The reference holds no golden vectors for this path (PARITY UNPINNED), so the oracle is pinned
by (1) an independent unbanded probability-space numpy forward/backward written differently on purpose
(helpers.full_matrix_reference), (2) invariants the algorithm must satisfy, (3) the committed golden
fixtures (test_golden.py).
"""
import os

import numpy as np
import pytest

from helpers import ROOT, cigar_spans, full_matrix_reference, load_model_arrays, oracle_hmm, orc, random_pair

LN2 = np.log(2.0)


@pytest.mark.parametrize("model", ["blasr_hmm_0.txt", "blasr_hmm_20.txt", "blasr_hmm_40.txt"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_full_band_matches_independent_numpy(model, seed):
    rng = np.random.default_rng(seed)
    T, E, _ = load_model_arrays(model)
    h = orc.make_hmm(T, E)
    X, Y, ops = random_pair(rng, int(rng.integers(5, 45)))
    seg = orc.plan(len(X), len(Y), ops, orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=10000))[0]
    assert seg["cells"] == (len(X) + 1) * (len(Y) + 1)
    r = orc.fb_f64(h, X, Y, seg["lo"], seg["n"])
    tot, post, _, _ = full_matrix_reference(T, E, X, Y)
    assert r["rc"] == 0
    assert r["total_ll"] == pytest.approx(np.log(tot), abs=1e-10)
    assert r["total_ll_bwd"] == pytest.approx(np.log(tot), abs=1e-10)
    ref = {(x, y): post[x, y] for x in range(len(X)) for y in range(len(Y)) if post[x, y] >= 0.01}
    got = {(int(x), int(y)): p for x, y, p in zip(r["px"], r["py"], r["pp"])}
    assert got.keys() == ref.keys()
    assert max(abs(ref[k] - got[k]) for k in ref) < 1e-11


def test_dense_stock_like_model_with_switch_transitions():
    """A model that uses all 15 transitions of the five-state cell update (short-gap switches non-zero)."""
    rng = np.random.default_rng(5)
    from nanopore_amd.hmm import stockHmm
    s = stockHmm()
    h = orc.make_hmm(s.transitions, s.emissions)
    X, Y, ops = random_pair(rng, 30)
    seg = orc.plan(len(X), len(Y), ops, orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=10000))[0]
    r = orc.fb_f64(h, X, Y, seg["lo"], seg["n"])
    tot, post, _, _ = full_matrix_reference(s.transitions, s.emissions, X, Y)
    assert r["total_ll"] == pytest.approx(np.log(tot), abs=1e-10)
    m = orc.fb_f32(h, X, Y, seg["lo"], seg["n"])
    assert (np.log2(m["tot_m"]) + m["tot_e"]) * LN2 == pytest.approx(np.log(tot), abs=1e-4)
    got = {(int(x), int(y)): p for x, y, p in zip(m["px"], m["py"], m["pp"])}
    for (x, y), p in got.items():
        assert abs(post[x, y] - p) < 1e-5


def test_ragged_ends_against_numpy():
    rng = np.random.default_rng(6)
    T, E, _ = load_model_arrays()
    Tm = T.reshape(5, 5)
    h = orc.make_hmm(T, E)
    X, Y, ops = random_pair(rng, 25)
    seg = orc.plan(len(X), len(Y), ops, orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=10000))[0]
    r = orc.fb_f64(h, X, Y, seg["lo"], seg["n"], ragged_start=1, ragged_end=1)
    start = np.array([0, 0, 0, 1.0, 1.0])
    end = np.array([Tm[0, 3], Tm[0, 3], Tm[0, 4], Tm[3, 3], Tm[4, 4]])
    tot, post, _, _ = full_matrix_reference(T, E, X, Y, start=start, end=end)
    assert r["total_ll"] == pytest.approx(np.log(tot), abs=1e-10)
    assert r["total_ll_bwd"] == pytest.approx(np.log(tot), abs=1e-10)


@pytest.mark.parametrize("seed", range(6))
def test_banded_invariants_and_fp32_mirror(seed):
    rng = np.random.default_rng(100 + seed)
    h = oracle_hmm()
    X, Y, ops = random_pair(rng, int(rng.integers(50, 600)), indel=0.15, max_indel=12)
    P = orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=int(rng.choice([20, 64, 100, 200])))
    seg = orc.plan(len(X), len(Y), ops, P)[0]
    r = orc.fb_f64(h, X, Y, seg["lo"], seg["n"])
    assert r["rc"] == 0
    assert r["total_ll_bwd"] == pytest.approx(r["total_ll"], rel=1e-12)      # F-total == B-total
    assert ((r["pp"] >= 0.01) & (r["pp"] <= 1.0 + 1e-9)).all()
    rowsum = np.bincount(r["py"], weights=r["pp"], minlength=len(Y))
    colsum = np.bincount(r["px"], weights=r["pp"], minlength=len(X))
    assert rowsum.max() <= 1.0 + 1e-9 and colsum.max() <= 1.0 + 1e-9          # a base pairs at most once
    m = orc.fb_f32(h, X, Y, seg["lo"], seg["n"])
    assert m["rc"] == 0
    ll32 = (np.log2(m["tot_m"]) + m["tot_e"]) * LN2
    assert ll32 == pytest.approx(r["total_ll"], rel=2e-6)
    d64 = {(int(x), int(y)): p for x, y, p in zip(r["px"], r["py"], r["pp"])}
    d32 = {(int(x), int(y)): float(p) for x, y, p in zip(m["px"], m["py"], m["pp"])}
    for k in set(d64) | set(d32):
        a, b = d64.get(k), d32.get(k)
        if a is None or b is None:
            assert abs((a if a is not None else b) - 0.01) < 1e-4
        else:
            assert abs(a - b) < 1e-4
    # renormalised log-probabilities: the tolerance north_star states (1e-4) on every live cell
    alive = np.isfinite(r["Fm"]) & (m["Fm_v"] > 0)
    lf = (np.log2(m["Fm_v"][alive].astype(np.float64)) + m["Fm_e"][alive]) * LN2
    assert np.abs(lf - r["Fm"][alive]).max() < 1e-4
    alive = np.isfinite(r["Bm"]) & (m["Bm_v"] > 0)
    lb = (np.log2(m["Bm_v"][alive].astype(np.float64)) + m["Bm_e"][alive]) * LN2
    assert np.abs(lb - r["Bm"][alive]).max() < 1e-4
    # dead cells agree
    assert ((m["Fm_v"] == 0) == ~np.isfinite(r["Fm"])).all()


def _check_band(seg, lX, lY):
    lo, n, D = seg["lo"].astype(np.int64), seg["n"].astype(np.int64), seg["D"]
    d = np.arange(D + 1)
    hi = lo + 2 * (n - 1)
    assert (n >= 1).all()
    assert ((lo - d) % 2 == 0).all()                                  # parity of the anti-diagonal
    assert (lo >= np.maximum(-d, d - 2 * lY)).all() and (hi <= np.minimum(d, 2 * lX - d)).all()  # inside lattice
    assert lo[0] == 0 and n[0] == 1 and lo[D] == lX - lY and n[D] == 1  # corners are in the band
    # every diagonal can be reached from the previous one by an x- or a y-step
    assert (np.maximum(lo[1:], lo[:-1] - 1) <= np.minimum(hi[1:], hi[:-1] + 1)).all()


@pytest.mark.parametrize("seed", range(8))
def test_band_construction_invariants(seed):
    rng = np.random.default_rng(200 + seed)
    X, Y, ops = random_pair(rng, int(rng.integers(20, 500)), indel=0.2, max_indel=int(rng.integers(1, 60)))
    for kw in (dict(band_mode=orc.BAND_FIXED, fixed_width=int(rng.integers(2, 80))),
               dict(band_mode=orc.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=int(rng.integers(0, 5)),
                    split_threshold=int(rng.integers(1, 40))),
               dict(band_mode=orc.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=3000)):
        segs = orc.plan(len(X), len(Y), ops, orc.make_params(**kw))
        assert segs[0]["xs"] == 0 and segs[0]["ys"] == 0 and segs[-1]["xe"] == len(X) and segs[-1]["ye"] == len(Y)
        assert segs[0]["ragged_start"] == 0 and segs[-1]["ragged_end"] == 0
        for a, b in zip(segs[:-1], segs[1:]):
            assert a["ragged_end"] == 1 and b["ragged_start"] == 1
            assert a["xe"] <= b["xs"] and a["ye"] <= b["ys"]                    # segments never overlap
        for s in segs:
            _check_band(s, s["xe"] - s["xs"], s["ye"] - s["ys"])
        if kw["band_mode"] == orc.BAND_FIXED:
            assert len(segs) == 1
            lo = segs[0]["lo"].astype(np.int64)
            assert (np.abs(np.diff(lo)) == 1).all()                           # a fixed-width band is a staircase


def test_anchor_band_stripe_width_and_split():
    # 200 matches, trim 0: a stripe of half-width diagonalExpansion around the main diagonal
    ops = [(0, 200)]
    s = orc.plan(200, 200, ops, orc.make_params(band_mode=orc.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=0))[0]
    mid = s["n"][100:300]
    assert set(mid.tolist()) == {11, 12}
    # a 50 x 40 unanchored rectangle with threshold 20 (area 2000 > 400) is cut: both sides keep min(gap/2, 20)
    ops = [(0, 30), (2, 50), (1, 40), (0, 30)]
    segs = orc.plan(110, 100, ops, orc.make_params(band_mode=orc.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=0,
                                                  split_threshold=20))
    assert len(segs) == 2
    assert (segs[0]["xe"], segs[0]["ye"]) == (30 + 20, 30 + 20)
    assert (segs[1]["xs"], segs[1]["ys"]) == (81 - 20, 71 - 20)
    # not global -> rejected (utils.py:381-382)
    with pytest.raises(ValueError):
        orc.plan(10, 10, [(0, 9)], orc.make_params())


@pytest.mark.parametrize("seed", range(5))
def test_mea_fenwick_equals_bruteforce_and_cigar_is_global(seed):
    rng = np.random.default_rng(300 + seed)
    h = oracle_hmm()
    X, Y, ops = random_pair(rng, int(rng.integers(30, 400)), indel=0.2, max_indel=8)
    P = orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=60)
    seg = orc.plan(len(X), len(Y), ops, P)[0]
    r = orc.fb_f64(h, X, Y, seg["lo"], seg["n"])
    for gg, mg in ((0.5, 0.0), (0.0, 0.0), (0.9, 0.3)):
        a, sa = orc.mea_cigar(len(X), len(Y), r["px"], r["py"], r["pp"], gg, mg)
        b, sb = orc.mea_cigar(len(X), len(Y), r["px"], r["py"], r["pp"], gg, mg, brute_force=True)
        assert a == b and sa == sb
        assert cigar_spans(a) == (len(X), len(Y))                               # utils.py:381-382 / :602
        assert all(k > 0 for _, k in a) and all(a[i][0] != a[i + 1][0] for i in range(len(a) - 1))
        assert 0.0 <= sa <= 1.0
    # no pairs at all: everything is unaligned
    assert orc.mea_cigar(7, 5, [], [], [])[0] == [(2, 7), (1, 5)]


def test_identical_sequences_realign_to_all_match_and_rescore():
    rng = np.random.default_rng(9)
    h = oracle_hmm()
    X = rng.integers(0, 4, size=300).astype(np.uint8)
    ops = [(0, 300)]
    P = orc.make_params(band_mode=orc.BAND_ANCHOR)
    r = orc.realign_read(h, P, X, X, ops)
    assert r["status"] == 0 and r["ops"] == [(0, 300)] and r["score"] > 0.95
    P2 = orc.make_params(band_mode=orc.BAND_ANCHOR, split_threshold=100, mode=orc.MODE_RESCORE_ORIGINAL)
    r2 = orc.realign_read(h, P2, X, X, ops)
    assert r2["ops"] == ops                                                    # alignmentUncertainty.py:51-52
    assert r2["score"] == pytest.approx(orc.rescore(ops, r2["px"], r2["py"], r2["pp"]))
    assert r2["score"] == pytest.approx(r["score"], abs=1e-6)
    # a guide that is wrong everywhere rescoring to ~0
    bad = [(1, 300), (2, 300)]
    r3 = orc.realign_read(h, orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=40, mode=orc.MODE_RESCORE_ORIGINAL), X, X, bad)
    assert r3["ops"] == bad and r3["score"] == 0.0


def test_unsupported_transition_rejected_by_mirror():
    T, E, _ = load_model_arrays()
    T = T.copy()
    T[3 * 5 + 1] = 0.01  # longGapX -> shortGapX is not part of the five-state cell update
    h = orc.make_hmm(T, E)
    m = orc.fb_f32(h, np.zeros(3, np.uint8), np.zeros(3, np.uint8), np.array([0, -1, -2, -1, 0, 1, 0], np.int32),
                   np.array([1, 2, 3, 3, 3, 2, 1], np.int32))
    assert m["rc"] == -4


def test_expectations_match_independent_numpy():
    """Baum-Welch expected counts against counts computed from the independent numpy forward/backward."""
    rng = np.random.default_rng(31)
    T, E, _ = load_model_arrays()
    h = orc.make_hmm(T, E)
    X, Y, ops = random_pair(rng, 30)
    X[7] = 4  # an N: contributes to transitions, not to emissions
    seg = orc.plan(len(X), len(Y), ops, orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=10000))[0]
    r = orc.expectations(h, X, Y, seg["lo"], seg["n"])
    tot, _, F, B = full_matrix_reference(T, E, X, Y)
    Tm = np.asarray(T).reshape(5, 5)
    Em = np.full((5, 5), 1.0 / 16.0)
    Em[:4, :4] = np.asarray(E[:16]).reshape(4, 4)
    Ex = {s: np.append(np.asarray(E[16 * s:16 * s + 16]).reshape(4, 4).sum(axis=1), 0.25) for s in range(5)}
    Ey = {s: np.append(np.asarray(E[16 * s:16 * s + 16]).reshape(4, 4).sum(axis=0), 0.25) for s in range(5)}
    Texp, Eexp = np.zeros((5, 5)), np.zeros(80)
    for x in range(len(X) + 1):
        for y in range(len(Y) + 1):
            if x > 0 and y > 0:
                w = F[x - 1, y - 1] * Tm[:, 0] * Em[X[x - 1], Y[y - 1]] * B[x, y, 0] / tot
                Texp[:, 0] += w
                if X[x - 1] < 4 and Y[y - 1] < 4:
                    Eexp[X[x - 1] * 4 + Y[y - 1]] += w.sum()
            if x > 0:
                for t in (1, 3):
                    w = F[x - 1, y] * Tm[:, t] * Ex[t][X[x - 1]] * B[x, y, t] / tot
                    Texp[:, t] += w
                    if X[x - 1] < 4:
                        Eexp[t * 16 + X[x - 1] * 4:t * 16 + X[x - 1] * 4 + 4] += 0.25 * w.sum()
            if y > 0:
                for t in (2, 4):
                    w = F[x, y - 1] * Tm[:, t] * Ey[t][Y[y - 1]] * B[x, y, t] / tot
                    Texp[:, t] += w
                    if Y[y - 1] < 4:
                        Eexp[t * 16 + Y[y - 1]:t * 16 + 16:4] += 0.25 * w.sum()
    assert r["rc"] == 0
    assert np.abs(r["T"] - Texp.reshape(-1)).max() < 1e-11
    assert np.abs(r["E"] - Eexp).max() < 1e-11
    assert r["total_ll"] == pytest.approx(np.log(tot), abs=1e-10)
    # every path has one transition per alignment column
    assert max(len(X), len(Y)) <= r["T"].sum() <= len(X) + len(Y)


def test_approximate_logadd_mode_is_a_measuring_stick_not_the_norm():
    """cPecan adds log-probabilities with a piecewise-cubic lookup [RECALLED].  The
    oracle offers it as a switch so that the distance "exact vs the reference's own approximation" can be put next to
    "GPU vs exact" (tools/logadd_risk.py).  Here: the recalled coefficients do approximate
    log(1 + e^t) (to 3e-4 on [0, 7.5], "max" beyond), the switch changes results by about that much and no more, and it
    switches back."""
    import math
    L = orc.lib()
    assert L.orc_get_logadd_kind() == orc.LOGADD_EXACT
    rng = np.random.default_rng(12)
    X, Y, g = random_pair(rng, 300)
    h = oracle_hmm()
    P = orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=60)
    exact = orc.realign_read(h, P, X, Y, g, precision=0)
    with orc.logadd_kind(orc.LOGADD_APPROX):
        assert L.orc_get_logadd_kind() == orc.LOGADD_APPROX
        worst = max(abs(L.orc_logadd(0.0, -t) - math.log1p(math.exp(-t))) for t in np.linspace(0.0, 7.49, 3000))
        assert 1e-5 < worst < 3e-4
        assert L.orc_logadd(-3.0, -11.0) == -3.0 and L.orc_logadd(float("-inf"), -2.0) == -2.0
        approx = orc.realign_read(h, P, X, Y, g, precision=0)
    assert L.orc_get_logadd_kind() == orc.LOGADD_EXACT
    assert orc.realign_read(h, P, X, Y, g, precision=0)["total_ll"] == exact["total_ll"]
    assert approx["total_ll"] != exact["total_ll"] and abs(approx["total_ll"] - exact["total_ll"]) < 3e-4 * 600
    de = {(int(a), int(b)): float(c) for a, b, c in zip(exact["px"], exact["py"], exact["pp"])}
    da = {(int(a), int(b)): float(c) for a, b, c in zip(approx["px"], approx["py"], approx["pp"])}
    diff = max(abs(de.get(k, 0.01) - da.get(k, 0.01)) for k in set(de) | set(da))
    assert 1e-6 < diff < 0.2   # two orders of magnitude more than fp32-vs-fp64 (1e-6, test_fp32_mirror_*), far from garbage


def test_oracle_is_clean_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """the oracle is the project's C code that every parity claim rests on; its own tests run once
    with -fsanitize=address,undefined (a subprocess: the sanitizer runtime has to be loaded before Python)."""
    import shutil
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(orc.__file__))
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    work = tmp_path / "oracle"
    shutil.copytree(here, str(work), ignore=shutil.ignore_patterns("*.so", "__pycache__", "_ref"))
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-std=gnu11", "-fPIC", "-fopenmp", "-ffp-contract=off", "-shared", "-o", str(work / "liboracle.so"),
                           str(work / "realign_oracle.c"), str(work / "realign_oracle_f32.c"), str(work / "realign_oracle_rs.c"), "-lm"])
    shutil.copy(str(work / "liboracle.so"), str(work / "liboracle_native.so"))
    for so in ("liboracle.so", "liboracle_native.so"):   # newer than the sources: oracle.build() keeps them
        os.utime(str(work / so))
    script = (
        "import sys, os; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)\n"  # the sanitized copy wins
        "import numpy as np\n"
        "from oracle import oracle as orc\n"
        "assert os.path.dirname(orc.__file__) == %r\n"
        "from helpers import oracle_hmm, random_pair\n"
        "rng = np.random.default_rng(3)\n"
        "h = oracle_hmm()\n"
        "for n, P in ((40, orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=20)), (300, orc.make_params(band_mode=orc.BAND_ANCHOR, constraint_trim=3, split_threshold=40)),\n"
        "             (200, orc.make_params(band_mode=orc.BAND_ANCHOR, mode=orc.MODE_RESCORE_ORIGINAL, split_threshold=100)), (0, orc.make_params())):\n"
        "    X, Y, g = random_pair(rng, n) if n else (np.zeros(0, np.uint8), np.zeros(0, np.uint8), [])\n"
        "    for prec, arith in ((0, None), (1, None), (1, [1] * 64)):\n"
        "        r = orc.realign_read(h, P, X, Y, g, precision=prec, seg_arith=arith)\n"
        "        assert r['status'] in (0, -1), r['status']\n"
        "    for seg in (orc.plan(len(X), len(Y), g, P) if n else []):\n"
        "        e = orc.expectations(h, X[seg['xs']:seg['xe']], Y[seg['ys']:seg['ye']], seg['lo'], seg['n'], seg['ragged_start'], seg['ragged_end'])\n"
        "        assert e['rc'] == 0\n"
        "print('SANITIZED OK')\n") % (ROOT, os.path.join(ROOT, "tests"), str(tmp_path), str(work))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "SANITIZED OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def _pairs(r):
    return {(int(a), int(b)): float(c) for a, b, c in zip(r["px"], r["py"], r["pp"])}


def test_row_scaled_mirror_equals_the_per_cell_mirror_and_tracks_fp64():
    """The two fp32 restatements of the device arithmetic (realign_oracle_f32.c: one exponent per cell; realign_oracle_rs.c:
    one per anti-diagonal row, the kernels of narrow bands) differ only by exact powers of two while every cell stays inside
    fp32's range relative to its row: same posterior bits, same cigars, same totals -- and both within 1e-4 of fp64."""
    h = oracle_hmm()
    rng = np.random.default_rng(41)
    for W, n, lmax, indel, mi in ((40, 6, 400, 0.12, 6), (100, 6, 1500, 0.2, 30), (200, 6, 2500, 0.2, 60), (250, 3, 1500, 0.25, 100)):
        P = orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=W)
        for _ in range(n):
            X, Y, g = random_pair(rng, int(rng.integers(30, lmax)), indel=indel, max_indel=mi)
            a = orc.realign_read(h, P, X, Y, g, precision=1)
            b = orc.realign_read(h, P, X, Y, g, precision=1, seg_arith=[1] * 16)
            c = orc.realign_read(h, P, X, Y, g, precision=0)
            assert a["status"] == b["status"] == c["status"] == 0
            pa, pb, pc = _pairs(a), _pairs(b), _pairs(c)
            assert pa.keys() == pb.keys() and all(np.float32(pa[k]) == np.float32(pb[k]) for k in pa)
            assert a["ops"] == b["ops"] and a["total_ll"] == b["total_ll"] and a["score"] == b["score"]
            for k in set(pb) | set(pc):
                u, v = pb.get(k), pc.get(k)
                assert abs((u if u is not None else 0.01) - (v if v is not None else 0.01)) < 1e-4
            assert b["total_ll"] == pytest.approx(c["total_ll"], rel=2e-6)


def test_row_scaled_mirror_keeps_a_long_indel_the_guide_does_not_have_and_says_so():
    """The hard case for one exponent per row: the guide is a plain diagonal, the truth has a long deletion followed by an
    equally long insertion inside the band, so for a stretch the alignment runs along the far edge of the band while the
    largest forward values of those rows belong to the diagonal it left -- up to ~270 binary orders apart where a gap base
    costs ~1.4 bit more than an aligned one.  With its rows' maxima near the top of fp32's range the row-scaled arithmetic
    still keeps those cells (posteriors within 1e-4 of fp64, the fp64 cigar, like the per-cell mirror), and its range
    certificate (rc = 1: a row with eF + eB - eTot >= NPR_RS_S_LIMIT) marks the reads for which that was not guaranteed --
    the device runs those again with a per-cell exponent -- while ordinary reads pass it."""
    h = oracle_hmm()
    rng = np.random.default_rng(43)
    flagged = {}
    for gap in (30, 60, 150, 260):
        core = rng.integers(0, 4, size=1200).astype(np.uint8)
        extra_ref = rng.integers(0, 4, size=gap).astype(np.uint8)
        extra_read = rng.integers(0, 4, size=gap).astype(np.uint8)
        X = np.concatenate([core[:400], extra_ref, core[400:800], core[800:]])   # reference has `gap` bases the read lacks ...
        Y = np.concatenate([core[:400], core[400:800], extra_read, core[800:]])  # ... and the read `gap` bases of its own 400 later
        g = [(0, len(X))]                                                        # the guide knows of neither
        P = orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=2 * gap + 40)
        a = orc.realign_read(h, P, X, Y, g, precision=1)
        b = orc.realign_read(h, P, X, Y, g, precision=1, seg_arith=[1])
        c = orc.realign_read(h, P, X, Y, g, precision=0)
        assert a["status"] == b["status"] == c["status"] == 0
        assert any(op == 2 and n >= gap - 5 for op, n in c["ops"]) and any(op == 1 and n >= gap - 5 for op, n in c["ops"])  # found
        assert b["ops"] == c["ops"] == a["ops"]
        pb, pc = _pairs(b), _pairs(c)
        for k in set(pb) | set(pc):
            u, v = pb.get(k), pc.get(k)
            assert abs((u if u is not None else 0.01) - (v if v is not None else 0.01)) < 1e-4, (gap, k, u, v)
        assert b["total_ll"] == pytest.approx(c["total_ll"], rel=2e-6)
        seg = orc.plan(len(X), len(Y), g, P)[0]
        flagged[gap] = orc.fb_f32(h, X, Y, seg["lo"], seg["n"], dense=False, arith=1)["rc"]
    assert flagged[30] == 0 and flagged[60] == 0 and flagged[260] == 1, flagged
    for _ in range(6):  # reads with the usual indels pass the certificate
        X, Y, g = random_pair(rng, int(rng.integers(200, 2000)), indel=0.2, max_indel=30)
        seg = orc.plan(len(X), len(Y), g, orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=200))[0]
        assert orc.fb_f32(h, X, Y, seg["lo"], seg["n"], dense=False, arith=1)["rc"] == 0
