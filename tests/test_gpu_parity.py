"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Two bars (DESIGN.md "Parity"):
  * bit-exact against the oracle's fp32 mirror of the device arithmetic (posteriors, totals, dense
    forward/backward values, cigars);
  * within 1e-4 of the double-precision log-space oracle on posteriors and renormalised log-probabilities.
"""
import numpy as np
import pytest

from nanopore_amd import _lib

from helpers import load_model_arrays, oracle_hmm, orc, random_pair, cigar_spans  # noqa: F401

pytestmark = pytest.mark.gpu

LN2 = np.log(2.0)


def _hmm_obj(name):
    from nanopore_amd.hmm import Hmm
    from helpers import MODEL_DIR
    import os
    return Hmm.loadHmm(os.path.join(MODEL_DIR, name))


def _pairs_dict(x, y, p):
    return {(int(a), int(b)): float(c) for a, b, c in zip(x, y, p)}


def _run_case(gpu_ctx, rng, n_reads, lmin, lmax, kw, model="blasr_hmm_0.txt", indel=0.12, max_indel=4):
    from nanopore_amd import realign as R
    gpu_ctx.set_hmm(_hmm_obj(model))
    h = oracle_hmm(model)
    refs, reads, guides, raw = [], [], [], []
    for _ in range(n_reads):
        X, Y, ops = random_pair(rng, int(rng.integers(lmin, lmax + 1)), indel=indel, max_indel=max_indel)
        raw.append((X, Y, ops))
        refs.append(bytes(b"ACGT"[c] for c in X))
        reads.append(bytes(b"ACGT"[c] for c in Y))
        guides.append(ops)
    out = gpu_ctx.realign(R.make_params(**kw), refs, reads, guides, want_pairs=True)
    okw = dict(kw)
    okw.pop("max_pairs_per_base", None)
    P = orc.make_params(**okw)
    for (X, Y, ops), g in zip(raw, out):
        m32 = orc.realign_read(h, P, X, Y, ops, precision=1, seg_arith=g["seg_arith"])
        m64 = orc.realign_read(h, P, X, Y, ops, precision=0)
        assert g["status"] == 0 and m32["status"] == 0 and m64["status"] == 0, (g["status"], m32["status"], m64["status"], len(X), len(Y), g.get("seg_arith"))
        assert g["cells"] == m32["cells"] == m64["cells"]
        # --- bit-exact against the fp32 mirror ---
        gp = _pairs_dict(g["x"], g["y"], g["p"])
        mp = _pairs_dict(m32["px"], m32["py"], m32["pp"].astype(np.float32))
        assert gp.keys() == mp.keys()
        assert all(np.float32(gp[k]) == np.float32(mp[k]) for k in gp), "posterior differs from fp32 mirror"
        assert g["ops"] == m32["ops"], "cigar differs from fp32 mirror"
        assert g["score"] == pytest.approx(m32["score"], abs=1e-12)
        assert g["loglik"] == pytest.approx(m32["total_ll"], rel=1e-12, abs=1e-9)
        # --- tolerance against the fp64 log-space oracle ---
        dp = _pairs_dict(m64["px"], m64["py"], m64["pp"])
        for k in set(gp) | set(dp):
            a, b = gp.get(k), dp.get(k)
            if a is None or b is None:
                # a pair may sit on either side of the 0.01 threshold
                assert abs((a if a is not None else b) - kw.get("posterior_threshold", 0.01)) < 1e-4
            else:
                assert abs(a - b) < 1e-4
        assert g["loglik"] == pytest.approx(m64["total_ll"], rel=2e-6)
        assert g["loglik_bwd"] == pytest.approx(g["loglik"], rel=2e-6)
        assert cigar_spans(g["ops"]) == (len(X), len(Y))
    return out


def test_fixed_band_small(gpu_ctx):
    rng = np.random.default_rng(11)
    _run_case(gpu_ctx, rng, 24, 5, 300, dict(band_mode=1, fixed_width=40))


def test_fixed_band_wider_than_wave(gpu_ctx):
    rng = np.random.default_rng(12)
    _run_case(gpu_ctx, rng, 8, 300, 600, dict(band_mode=1, fixed_width=200))
    _run_case(gpu_ctx, rng, 4, 300, 500, dict(band_mode=1, fixed_width=700))  # generic kernel (> 256 cells)
    _run_case(gpu_ctx, rng, 4, 400, 700, dict(band_mode=1, fixed_width=400))  # register kernel, 4 cells per lane


def test_generic_and_register_kernels_agree(gpu_ctx, monkeypatch):
    """The LDS-ring kernel (any band) and the register staircase kernel share the cell arithmetic: forcing
    the generic kernel on a staircase batch must reproduce the same bits."""
    rng = np.random.default_rng(16)
    st = np.random.default_rng(16).bit_generator.state
    a = _run_case(gpu_ctx, rng, 6, 200, 900, dict(band_mode=1, fixed_width=100), indel=0.2, max_indel=30)
    gpu_ctx.set_option(_lib.OPTIONS["kernel"], 1)
    rng2 = np.random.default_rng(16)
    b = _run_case(gpu_ctx, rng2, 6, 200, 900, dict(band_mode=1, fixed_width=100), indel=0.2, max_indel=30)
    for u, v in zip(a, b):
        assert u["ops"] == v["ops"] and np.array_equal(u["p"], v["p"]) and u["loglik"] == v["loglik"]


def test_anchor_band_with_splits(gpu_ctx):
    rng = np.random.default_rng(13)
    _run_case(gpu_ctx, rng, 16, 50, 500, dict(band_mode=0, diagonal_expansion=10, constraint_trim=2,
                                             split_threshold=12), indel=0.2, max_indel=40)
    _run_case(gpu_ctx, rng, 8, 200, 800, dict(band_mode=0, diagonal_expansion=10, constraint_trim=14,
                                            split_threshold=3000))


def test_models_20_40_and_stock(gpu_ctx):
    rng = np.random.default_rng(14)
    for m in ("blasr_hmm_20.txt", "blasr_hmm_40.txt"):
        _run_case(gpu_ctx, rng, 6, 100, 300, dict(band_mode=1, fixed_width=60), model=m)


def test_dense_forward_backward_bit_exact(gpu_ctx):
    from nanopore_amd import realign as R
    rng = np.random.default_rng(15)
    gpu_ctx.set_hmm(_hmm_obj("blasr_hmm_0.txt"))
    h = oracle_hmm()
    X, Y, ops = random_pair(rng, 400)
    kw = dict(band_mode=1, fixed_width=100)
    seg = orc.plan(len(X), len(Y), ops, orc.make_params(**kw))[0]
    b = gpu_ctx.stage(R.make_params(**kw), [bytes(b"ACGT"[c] for c in X)], [bytes(b"ACGT"[c] for c in Y)], [ops])
    fv, fe, bv, be = b.dense(0, seg["cells"])
    b.close()
    m = orc.fb_f32(h, X, Y, seg["lo"], seg["n"])
    assert (fe == m["Fm_e"]).all() and (fv == m["Fm_v"]).all()
    assert (be == m["Bm_e"]).all() and (bv == m["Bm_v"]).all()
    # renormalised log-probabilities vs the fp64 oracle: subtract the per-read maximum (an offset)
    d = orc.fb_f64(h, X, Y, seg["lo"], seg["n"])
    alive = np.isfinite(d["Fm"]) & (fv > 0)
    lf = (np.log2(fv[alive].astype(np.float64)) + fe[alive]) * LN2
    assert np.abs((lf - d["Fm"][alive])).max() < 1e-4
    aliveb = np.isfinite(d["Bm"]) & (bv > 0)
    lb = (np.log2(bv[aliveb].astype(np.float64)) + be[aliveb]) * LN2
    assert np.abs((lb - d["Bm"][aliveb])).max() < 1e-4


@pytest.mark.parametrize("L,W,indel", [(400, 100, 0.1), (1500, 200, 0.2), (700, 40, 0.15), (900, 400, 0.2)], ids=["w100", "w200", "w40", "w400"])
def test_row_scaled_forward_rows_against_the_fp64_oracle(gpu_ctx, L, W, indel):
    """The rows k_dp_rs itself stores (one exponent per anti-diagonal row of the wavefront, renormalised every 16 rows), read out of
    its scratch: renormalised log-probabilities within the 1e-4 north_star states on every cell that carries posterior mass of
    2^-60 or more.  Below that a cell may have been flushed to zero (its forward value lies more than ~211 binary orders under
    its rows' maximum): then -- and only then -- the device holds 0 where the oracle holds a finite value, and the range certificate
    has bounded what such cells can carry.  The three frame classes, drifting bands (rebases)."""
    from nanopore_amd import realign as R
    rng = np.random.default_rng(151 + W)
    gpu_ctx.set_hmm(_hmm_obj("blasr_hmm_0.txt"))
    h = oracle_hmm()
    X, Y, ops = random_pair(rng, L, indel=indel, max_indel=30)
    kw = dict(band_mode=1, fixed_width=W)
    seg = orc.plan(len(X), len(Y), ops, orc.make_params(**kw))[0]
    b = gpu_ctx.stage(R.make_params(**kw), [bytes(b"ACGT"[c] for c in X)], [bytes(b"ACGT"[c] for c in Y)], [ops])
    tasks, _ = b.class_stats()
    assert tasks[15:18].sum() + tasks[12:15].sum() == 1  # a row-scaled class took it (k_dp_rs, or its sweeps on two wavefronts: the same rows)
    b.run()
    assert b.segment_arith()[1][0] == 1                  # ... and kept its range certificate
    fv, fe = b.rs_forward(0, seg["cells"])
    b.close()
    d = orc.fb_f64(h, X, Y, seg["lo"], seg["n"])
    post = d["Fm"] + d["Bm"]                             # log of forward x backward of the match state: total + log posterior
    tot = d["total_ll"]
    heavy = np.isfinite(post) & (post >= tot - 60 * LN2)
    assert heavy.sum() > L and (fv[heavy] > 0).all()    # nothing that matters was flushed
    lf = (np.log2(fv[heavy].astype(np.float64)) + fe[heavy]) * LN2
    assert np.abs(lf - d["Fm"][heavy]).max() < 1e-4
    # ... and everywhere else a positive value is as accurate; a zero is a flushed cell or one outside every path
    rest = np.isfinite(d["Fm"]) & (fv > 0) & ~heavy
    if rest.any():
        lr = (np.log2(fv[rest].astype(np.float64)) + fe[rest]) * LN2
        assert np.abs(lr - d["Fm"][rest]).max() < 1e-3  # (denormals: fewer bits)
    flushed = np.isfinite(d["Fm"]) & (fv == 0)
    assert not (flushed & heavy).any()


def test_edge_cases(gpu_ctx):
    from nanopore_amd import realign as R
    gpu_ctx.set_hmm(_hmm_obj("blasr_hmm_0.txt"))
    P = R.make_params(band_mode=1, fixed_width=20)
    # single base, pure insertion / deletion guides, N bases, an invalid (non-global) guide
    refs = [b"A", b"ACGT", b"ACGTNNACGT", b"ACGTACGT", b"ACGT"]
    reads = [b"A", b"ACGTTT", b"ACGTNNACGT", b"ACG", b"ACGT"]
    guides = [[(0, 1)], [(0, 4), (1, 2)], [(0, 10)], [(0, 3), (2, 5)], [(0, 3)]]
    out = gpu_ctx.realign(P, refs, reads, guides)
    assert [o["status"] for o in out[:4]] == [0, 0, 0, 0]
    assert out[4]["status"] == -1  # guide does not span the sequences: per-read error, batch survives
    for o, r, q in zip(out[:4], refs, reads):
        assert cigar_spans(o["ops"]) == (len(r), len(q))
    assert out[0]["ops"] == [(0, 1)]
    # empty batch
    assert gpu_ctx.realign(P, [], [], []) == []


@pytest.mark.timeout(300)
def test_posterior_capacity_overflow_is_reported_and_retried(gpu_ctx):
    """A sparse posterior list that does not fit its capacity is a per-read NPR_ERR_CAPACITY from the C ABI; the
    Python layer re-runs such reads with a larger capacity and ends with the same answer."""
    from nanopore_amd import realign as R
    rng = np.random.default_rng(17)
    gpu_ctx.set_hmm(_hmm_obj("blasr_hmm_0.txt"))
    cases = [random_pair(rng, 600) for _ in range(3)]
    refs = [bytes(b"ACGT"[c] for c in X) for X, _, _ in cases]
    reads = [bytes(b"ACGT"[c] for c in Y) for _, Y, _ in cases]
    guides = [g for _, _, g in cases]
    tight = R.make_params(band_mode=1, fixed_width=100, max_pairs_per_base=1)
    b = gpu_ctx.stage(tight, refs, reads, guides)
    b.run()
    b.finish()
    assert (b.results()["status"] == -3).all()       # ~1.8 pairs per base do not fit 1 per base
    b.close()
    a = gpu_ctx.realign(tight, refs, reads, guides, want_pairs=True)
    c = gpu_ctx.realign(R.make_params(band_mode=1, fixed_width=100), refs, reads, guides, want_pairs=True)
    for u, v in zip(a, c):
        assert u["status"] == 0 and u["ops"] == v["ops"] and np.array_equal(u["p"], v["p"])
    # A batch in which SOME lists overflow: the device MEA stage walks every task's list, so a list that overflowed must still hold
    # posteriors (k_dp_mid_rs's two wavefronts fill it from both ends with candidates that are only rescaled when it did not overflow --
    # round 5: a chain kernel fed such values did not come back), and the reads that fit are not disturbed by their neighbours.
    cases = [random_pair(rng, int(rng.integers(200, 1500)), indel=float(rng.choice([0.05, 0.3])), max_indel=12) for _ in range(160)]
    refs = [bytes(b"ACGT"[c] for c in X) for X, _, _ in cases]
    reads = [bytes(b"ACGT"[c] for c in Y) for _, Y, _ in cases]
    guides = [g for _, _, g in cases]
    b = gpu_ctx.stage(R.make_params(band_mode=1, fixed_width=100, max_pairs_per_base=2), refs, reads, guides)
    b.run()
    b.finish()
    st = b.results()["status"].copy()
    off, ops = b.ops()
    b.close()
    assert (st == -3).sum() >= 20 and (st == 0).sum() >= 20 and set(st.tolist()) <= {0, -3}
    roomy = gpu_ctx.realign(R.make_params(band_mode=1, fixed_width=100), refs, reads, guides)
    for i in np.nonzero(st == 0)[0]:
        assert [tuple(o) for o in ops[off[i]:off[i + 1]].tolist()] == [tuple(o) for o in roomy[i]["ops"]]


@pytest.mark.timeout(120)
def test_finish_stages_refuse_malformed_pair_lists_in_bounded_time(gpu_ctx):
    """The MEA stage against input no DP kernel should ever produce (round 5 lost 25 GPU-minutes to a chain kernel that was fed candidates instead
    of posteriors): pair lists overwritten on the device through the test hook npr_batch_debug_set_pairs -- out of order, the same pair twice,
    NaN / negative / above-1 values, coordinates outside the read, a list marked as overflowed -- must come back as a PER-READ status within
    a second, the reads beside them untouched; and the public host stage npr_mea_cigar refuses the same lists."""
    import time
    from nanopore_amd import realign as R
    rng = np.random.default_rng(29)
    gpu_ctx.set_hmm(_hmm_obj("blasr_hmm_0.txt"))
    cases = [random_pair(rng, 500) for _ in range(12)]
    refs = [bytes(b"ACGT"[c] for c in X) for X, _, _ in cases]
    reads = [bytes(b"ACGT"[c] for c in Y) for _, Y, _ in cases]
    guides = [g for _, _, g in cases]
    P = R.make_params(band_mode=1, fixed_width=100)
    good = gpu_ctx.realign(P, refs, reads, guides)
    n = 400
    diag = np.arange(n, dtype=np.int32)
    ones = np.full(n, 0.9, np.float32)
    nan = ones.copy(); nan[::7] = np.nan
    neg = ones.copy(); neg[5] = -0.5
    big = ones.copy(); big[11] = 3.0e9
    inf = ones.copy(); inf[3] = np.inf
    bad_lists = {
        0: (diag[::-1].copy(), diag[::-1].copy(), ones),                       # descending
        1: (np.repeat(diag[: n // 2], 2), np.repeat(diag[: n // 2], 2), ones),  # every pair twice
        2: (diag, diag, nan),
        3: (diag, diag, neg),
        4: (diag, diag, big),
        5: (diag, diag, inf),
        6: (diag + 100000, diag, ones),                                        # outside the reference
        7: (rng.permutation(diag).astype(np.int32), rng.permutation(diag).astype(np.int32), ones),  # no order at all
    }
    for host_mea in (0, 1):
        gpu_ctx.set_option(_lib.OPTIONS["host_mea"], host_mea)
        b = gpu_ctx.stage(P, refs, reads, guides)
        b.run()
        for r, (x, y, p) in bad_lists.items():
            b.debug_set_pairs(r, x, y, p)
        b.debug_set_pairs(8, diag, diag, ones, task_status=-3)                   # a list its kernel reported as overflowed: not read at all
        t0 = time.perf_counter()
        b.finish()
        dt = time.perf_counter() - t0
        st = b.results()["status"].copy()
        off, ops = b.ops()
        b.close()
        assert dt < 1.0, dt
        # values that are no probabilities and coordinates outside the read are refused by both stages; the device stage sorts whatever order it is
        # given (its pairs come unordered from the DP kernels) and takes a repeated pair as two pairs, the host stage gets its lists sorted
        for r in (2, 3, 4, 5, 6):
            assert st[r] == -1, (host_mea, r, st)
        assert st[8] == -3
        for r in (0, 1, 7):
            assert st[r] in (0, -1)
        for r in range(9, len(cases)):
            assert st[r] == 0 and [tuple(o) for o in ops[off[r]:off[r + 1]].tolist()] == [tuple(o) for o in good[r]["ops"]]
    gpu_ctx.set_option(_lib.OPTIONS["host_mea"], 0)
    # the public host stage on the same lists: it sorts what it is given (any order is fine), and refuses the rest
    want = R.mea_cigar(600, 600, diag, diag, ones)
    for r in (0, 7):
        x, y, p = bad_lists[r]
        if r == 0:
            assert R.mea_cigar(600, 600, x, y, p) == want
        else:
            R.mea_cigar(600, 600, x, y, p)
    for r in (1, 2, 3, 4, 5, 6):
        x, y, p = bad_lists[r]
        with pytest.raises(R.NprError):
            R.mea_cigar(600, 600, x, y, p)


def test_device_mea_matches_host_stage(gpu_ctx, monkeypatch):
    """The chain + cigar stage runs on the device in realign mode (npr_mea.hip) and on the host in the other modes,
    for hand-made pair lists (npr_mea_cigar) and with NPR_OPT_HOST_MEA: same integers, so same ops and same scores --
    over gapGamma / matchGamma settings, reads of several segments, N bases, reads without any pair above matchGamma,
    and with the pairs fetched afterwards."""
    from nanopore_amd import realign as R
    rng = np.random.default_rng(23)
    gpu_ctx.set_hmm(_hmm_obj("blasr_hmm_0.txt"))
    cases = [random_pair(rng, int(rng.integers(30, 2500)), indel=0.2, max_indel=40) for _ in range(40)]
    cases += [random_pair(rng, 1, indel=0.0), random_pair(rng, 3000, indel=0.3, max_indel=150)]
    cases[1][0][5:40] = 4
    refs = [bytes(b"ACGTN"[c] for c in X) for X, _, _ in cases]
    reads = [bytes(b"ACGTN"[c] for c in Y) for _, Y, _ in cases]
    guides = [g for _, _, g in cases]
    for kw in (dict(band_mode=1, fixed_width=100), dict(band_mode=1, fixed_width=100, gap_gamma=0.0),
               dict(band_mode=1, fixed_width=64, gap_gamma=0.9, match_gamma=0.3), dict(band_mode=1, fixed_width=700),
               dict(band_mode=0, split_threshold=40, constraint_trim=3), dict(band_mode=0, match_gamma=0.95),
               dict(band_mode=0, gap_gamma=0.2, match_gamma=-0.1)):
        got = {}
        # device_ring: the general (LDS-ring) chain kernel for every read and the global-memory sort kernels (the
        # variants long spans and far-reaching pairs fall back to)
        for where in ("device", "device_ring", "host"):
            for k in ("host_mea", "mea_ring_only", "mea_global_sort"):
                gpu_ctx.set_option(_lib.OPTIONS[k], 0)
            if where == "host":
                gpu_ctx.set_option(_lib.OPTIONS["host_mea"], 1)
            elif where == "device_ring":
                gpu_ctx.set_option(_lib.OPTIONS["mea_ring_only"], 1)
                gpu_ctx.set_option(_lib.OPTIONS["mea_global_sort"], 1)
            got[where] = gpu_ctx.realign(R.make_params(**kw), refs, reads, guides, want_pairs=(where == "device"))
        for k in ("host_mea", "mea_ring_only", "mea_global_sort"):
            gpu_ctx.set_option(_lib.OPTIONS[k], 0)
        for u, t, v in zip(got["device"], got["device_ring"], got["host"]):
            assert u["status"] == t["status"] == v["status"] == 0
            assert u["ops"] == v["ops"] and u["score"] == v["score"] and u["n_pairs"] == v["n_pairs"], kw
            assert t["ops"] == v["ops"] and t["score"] == v["score"], kw
            assert len(u["p"]) == u["n_pairs"]


def test_rescore_and_all_posteriors_modes_finish_on_the_device(gpu_ctx):
    """The call-site modes of the posterior consumers (nanopore/analyses/alignmentUncertainty.py:41 -- the analysis every experiment runs by
    default, pipeline.py:81 --, marginAlignSnpCaller.py:136-146) no longer copy the pairs out to be finished (round 5).
    NPR_MODE_RESCORE_ORIGINAL: the guide's M columns looked up where the pairs lie, summed in fixed point -- the SAME double as the host
    stage's walk over the sorted pair list (NPR_OPT_HOST_MEA = 1), and as the oracle's rescore of the fp32 mirror's pairs; the ops are the
    guide's.  NPR_MODE_ALL_POSTERIORS: the realign mode's device chain; the pairs come when asked for and are the host stage's.
    Guides with long indels (runs that start anywhere), N bases, a read of one base, anchors with the analyses' split 100, a fixed band."""
    from nanopore_amd import realign as R
    rng = np.random.default_rng(71)
    gpu_ctx.set_hmm(_hmm_obj("blasr_hmm_0.txt"))
    h = oracle_hmm()
    cases = [random_pair(rng, int(rng.integers(30, 2500)), indel=0.2, max_indel=40) for _ in range(30)]
    cases += [random_pair(rng, 1, indel=0.0), random_pair(rng, 3000, indel=0.3, max_indel=150)]
    cases[2][0][5:40] = 4
    refs = [bytes(b"ACGTN"[c] for c in X) for X, _, _ in cases]
    reads = [bytes(b"ACGTN"[c] for c in Y) for _, Y, _ in cases]
    guides = [g for _, _, g in cases]
    for kw in (dict(band_mode=0, diagonal_expansion=10, constraint_trim=14, split_threshold=100), dict(band_mode=1, fixed_width=100)):
        for mode in (R.MODE_RESCORE_ORIGINAL, R.MODE_ALL_POSTERIORS):
            got = {}
            for where in ("device", "host"):
                gpu_ctx.set_option(_lib.OPTIONS["host_mea"], int(where == "host"))
                got[where] = gpu_ctx.realign(R.make_params(mode=mode, **kw), refs, reads, guides, want_pairs=True)
            gpu_ctx.set_option(_lib.OPTIONS["host_mea"], 0)
            for (X, Y, ops), u, v in zip(cases, got["device"], got["host"]):
                assert u["status"] == v["status"] == 0
                assert u["ops"] == v["ops"] and u["score"] == v["score"] and u["n_pairs"] == v["n_pairs"], (kw, mode)
                assert np.array_equal(u["x"], v["x"]) and np.array_equal(u["y"], v["y"]) and np.array_equal(u["p"].view(np.uint32), v["p"].view(np.uint32))
                if mode == R.MODE_RESCORE_ORIGINAL:
                    assert u["ops"] == [(o, l) for o, l in ops if l > 0]
                    assert u["score"] == orc.rescore(np.asarray(ops, dtype=np.int32), u["x"], u["y"], u["p"])
                    m32 = orc.realign_read(h, orc.make_params(mode=orc.MODE_RESCORE_ORIGINAL, **kw), X, Y, ops, precision=1, seg_arith=u["seg_arith"])
                    assert u["score"] == m32["score"]


def test_device_mea_long_spans(gpu_ctx, monkeypatch):
    """Reads longer than the 16 k positions the in-LDS sort holds: the device chain takes its tables through HBM
    (k_mea_count / k_mea_scan / k_mea_scatter) without being told to; same ops and scores as the host stage."""
    from nanopore_amd import realign as R
    rng = np.random.default_rng(29)
    gpu_ctx.set_hmm(_hmm_obj("blasr_hmm_0.txt"))
    cases = [random_pair(rng, n, indel=0.15, max_indel=25) for n in (17000, 300, 21000)]
    refs = [bytes(b"ACGT"[c] for c in X) for X, _, _ in cases]
    reads = [bytes(b"ACGT"[c] for c in Y) for _, Y, _ in cases]
    guides = [g for _, _, g in cases]
    P = R.make_params(band_mode=1, fixed_width=80)
    dev = gpu_ctx.realign(P, refs, reads, guides)
    gpu_ctx.set_option(_lib.OPTIONS["host_mea"], 1)
    host = gpu_ctx.realign(P, refs, reads, guides)
    gpu_ctx.set_option(_lib.OPTIONS["host_mea"], 0)
    for u, v, (X, Y, _) in zip(dev, host, cases):
        assert u["status"] == v["status"] == 0 and u["ops"] == v["ops"] and u["score"] == v["score"]
        assert cigar_spans(u["ops"]) == (len(X), len(Y))
    # The packed cigars cross PCIe as 16-bit words when no run of the batch needs more than 14 bits (round 5) and as whole words
    # otherwise: both forms of the same batch, and a batch with runs of 17 000 and 21 000 (matchGamma 1 keeps no pair: a read's cigar
    # is its length inserted and its reference deleted).
    gpu_ctx.set_option(_lib.OPTIONS["mea_wide_ops"], 1)
    wide = gpu_ctx.realign(P, refs, reads, guides)
    gpu_ctx.set_option(_lib.OPTIONS["mea_wide_ops"], 0)
    assert [u["ops"] for u in wide] == [u["ops"] for u in dev]
    P1 = R.make_params(band_mode=1, fixed_width=80, match_gamma=1.0)
    dev = gpu_ctx.realign(P1, refs, reads, guides)
    gpu_ctx.set_option(_lib.OPTIONS["host_mea"], 1)
    host = gpu_ctx.realign(P1, refs, reads, guides)
    gpu_ctx.set_option(_lib.OPTIONS["host_mea"], 0)
    for u, v, (X, Y, _) in zip(dev, host, cases):
        assert u["status"] == v["status"] == 0 and u["ops"] == v["ops"] and u["score"] == v["score"] == 0.0
        assert sorted(u["ops"]) == sorted([(_lib.OP_I, len(Y)), (_lib.OP_D, len(X))])


def test_base_dependent_gap_emissions(gpu_ctx):
    """Gap-state emissions that depend on the base (and the flat N emission next to them): every shipped model is
    flat there, so this is the only place the per-base gap tables are exercised."""
    from nanopore_amd import realign as R
    from nanopore_amd.hmm import Hmm
    rng = np.random.default_rng(18)
    T, E, _ = load_model_arrays()
    E = E.copy()
    for s in range(1, 5):  # non-uniform gap emissions, still normalised
        blk = rng.random((4, 4)) + 0.2
        E[16 * s:16 * s + 16] = (blk / blk.sum()).reshape(-1)
    hm = Hmm()
    hm.transitions, hm.emissions = [float(v) for v in T], [float(v) for v in E]
    gpu_ctx.set_hmm(hm)
    h = orc.make_hmm(T, E)
    cases = [random_pair(rng, int(rng.integers(100, 500))) for _ in range(5)]
    cases[0][0][20:24] = 4  # N bases take the flat N emission
    refs = [bytes(b"ACGTN"[c] for c in X) for X, _, _ in cases]
    reads = [bytes(b"ACGTN"[c] for c in Y) for _, Y, _ in cases]
    for W in (60, 200):
        P = R.make_params(band_mode=1, fixed_width=W)
        out = gpu_ctx.realign(P, refs, reads, [g for _, _, g in cases], want_pairs=True)
        for (X, Y, g), o in zip(cases, out):
            m = orc.realign_read(h, orc.make_params(band_mode=1, fixed_width=W), X, Y, g, precision=1, seg_arith=o["seg_arith"])
            assert o["status"] == 0 and o["ops"] == m["ops"]
            order = np.lexsort((m["py"], m["px"]))
            assert np.array_equal(o["p"], m["pp"].astype(np.float32)[order])
    gpu_ctx.set_hmm(_hmm_obj("blasr_hmm_0.txt"))


def test_row_scaled_task_without_its_range_certificate_runs_again_per_cell(gpu_ctx):
    """k_dp_rs keeps one exponent per anti-diagonal row; a task with a row whose alignment lies too far below the row's largest
    forward and backward values to guarantee that nothing was flushed (NPR_RS_S_LIMIT, npr_device.h) is run again by
    npr_batch_run with the per-cell-exponent kernel.  A read with a 235-base deletion and, 400 bases on, a 235-base insertion
    the guide knows nothing of is such a task; its neighbours in the batch are not.  Either way the results are those of
    the matching CPU restatement bit for bit, and within 1e-4 of fp64."""
    from nanopore_amd import realign as R
    gpu_ctx.set_hmm(_hmm_obj("blasr_hmm_0.txt"))
    h = oracle_hmm()
    rng = np.random.default_rng(47)
    cases = []
    for gap in (235, 60):
        core = rng.integers(0, 4, size=1200).astype(np.uint8)
        X = np.concatenate([core[:400], rng.integers(0, 4, size=gap).astype(np.uint8), core[400:800], core[800:]])
        Y = np.concatenate([core[:400], core[400:800], rng.integers(0, 4, size=gap).astype(np.uint8), core[800:]])
        cases.append((X, Y, [(0, len(X))]))
    cases += [random_pair(rng, int(rng.integers(300, 1500)), indel=0.15, max_indel=20) for _ in range(6)]
    kw = dict(band_mode=1, fixed_width=500)  # 251 cells per anti-diagonal: the widest one-wavefront class
    out = gpu_ctx.realign(R.make_params(**kw), [bytes(b"ACGT"[c] for c in X) for X, _, _ in cases],
                          [bytes(b"ACGT"[c] for c in Y) for _, Y, _ in cases], [g for _, _, g in cases], want_pairs=True)
    assert out[0]["seg_arith"] == [0] and all(o["seg_arith"] == [1] for o in out[1:])
    P = orc.make_params(**kw)
    for (X, Y, g), o in zip(cases, out):
        m32 = orc.realign_read(h, P, X, Y, g, precision=1, seg_arith=o["seg_arith"])
        m64 = orc.realign_read(h, P, X, Y, g, precision=0)
        assert o["status"] == 0 and o["ops"] == m32["ops"] == m64["ops"]
        gp, mp, dp = _pairs_dict(o["x"], o["y"], o["p"]), _pairs_dict(m32["px"], m32["py"], m32["pp"].astype(np.float32)), _pairs_dict(m64["px"], m64["py"], m64["pp"])
        assert gp.keys() == mp.keys() and all(np.float32(gp[k]) == np.float32(mp[k]) for k in gp)
        for k in set(gp) | set(dp):
            a, b = gp.get(k), dp.get(k)
            assert abs((a if a is not None else 0.01) - (b if b is not None else 0.01)) < 1e-4
        assert o["loglik"] == pytest.approx(m64["total_ll"], rel=2e-6)
    assert any(op == 2 and n >= 230 for op, n in out[0]["ops"]) and any(op == 1 and n >= 230 for op, n in out[0]["ops"])


def test_band_without_probability_is_reported_by_the_per_cell_kernel(gpu_ctx):
    """A row-scaled task whose forward sweep arrives at the end corner with nothing cannot tell a band that carries no
    probability from one whose probability fell out of a row's range, so it too runs again with a per-cell exponent
    (npr_device.h TASK_RERUN): NPR_ERR_ZERO_PROB then comes from the kernel without a range limit.  A model without gaps and
    without mismatches makes a read with one substitution such a band; its twin without the substitution aligns."""
    from nanopore_amd import realign as R
    from nanopore_amd.hmm import Hmm
    T = np.zeros(25)
    T[0] = 1.0                                   # match -> match only
    E = np.zeros(80)
    for s in range(5):
        for b in range(4):
            E[16 * s + 5 * b] = 0.25             # every state emits identical bases only
    hm = Hmm()
    hm.transitions, hm.emissions = [float(v) for v in T], [float(v) for v in E]
    gpu_ctx.set_hmm(hm)
    rng = np.random.default_rng(53)
    X = rng.integers(0, 4, size=300).astype(np.uint8)
    Y = X.copy()
    Y[150] = (Y[150] + 1) % 4
    refs = [bytes(b"ACGT"[c] for c in X)] * 2
    reads = [bytes(b"ACGT"[c] for c in Y), bytes(b"ACGT"[c] for c in X)]
    out = gpu_ctx.realign(R.make_params(band_mode=1, fixed_width=40), refs, reads, [[(0, 300)], [(0, 300)]])
    assert out[0]["status"] == -2 and out[0]["seg_arith"] == [0]     # NPR_ERR_ZERO_PROB, said by k_dp_stair
    assert out[1]["status"] == 0 and out[1]["ops"] == [(0, 300)] and out[1]["seg_arith"] == [1]
    gpu_ctx.set_hmm(_hmm_obj("blasr_hmm_0.txt"))


def test_row_scaled_sweeps_in_blocks_of_sixteen_anti_diagonals(gpu_ctx):
    """k_dp_rs runs both sweeps in blocks of NPR_RS_K = 16 anti-diagonals (stream windows looked after once per block, the
    renormalising row the last of its block): every number of anti-diagonals from 2 to 70 -- shorter than a block, exactly one or
    several, one more, one less, odd and even -- and reads whose bands drift across several refills of the base streams, in the
    three frame classes.  Same bits as the mirror, which knows nothing of blocks."""
    from nanopore_amd import realign as R
    rng = np.random.default_rng(77)
    gpu_ctx.set_hmm(_hmm_obj("blasr_hmm_0.txt"))
    h = oracle_hmm("blasr_hmm_0.txt")
    kw = dict(band_mode=1, fixed_width=40)
    raw = []
    for D in range(2, 71):  # lX + lY = D exactly: a copy with substitutions, one inserted base when D is odd
        lx = D // 2
        X = rng.integers(0, 4, size=lx).astype(np.uint8)
        Y = np.where(rng.random(lx) < 0.1, (X + 1) % 4, X).astype(np.uint8)
        ops = [(0, lx)]
        if D & 1:
            Y, ops = np.append(Y, np.uint8(rng.integers(0, 4))), [(0, lx), (1, 1)]
        raw.append((X, Y, ops))
    for lx in range(1, 36):
        X, Y, ops = random_pair(rng, lx, indel=0.15, max_indel=2)
        if len(Y):
            raw.append((X, Y, ops))
    out = gpu_ctx.realign(R.make_params(**kw), [bytes(b"ACGT"[c] for c in X) for X, _, _ in raw],
                          [bytes(b"ACGT"[c] for c in Y) for _, Y, _ in raw], [g for _, _, g in raw], want_pairs=True)
    P = orc.make_params(**kw)
    n_rs = 0
    for (X, Y, ops), g in zip(raw, out):
        m = orc.realign_read(h, P, X, Y, ops, precision=1, seg_arith=g["seg_arith"])
        n_rs += g["seg_arith"] == [1]
        assert g["status"] == 0 and g["ops"] == m["ops"]
        gp, mp = _pairs_dict(g["x"], g["y"], g["p"]), _pairs_dict(m["px"], m["py"], m["pp"].astype(np.float32))
        assert gp.keys() == mp.keys() and all(np.float32(gp[k]) == np.float32(mp[k]) for k in gp), (len(X), len(Y))
    assert n_rs > len(raw) // 2
    # long drifting bands: many refills of both streams in both sweeps, rebases in both directions
    _run_case(gpu_ctx, rng, 6, 2000, 4000, dict(band_mode=1, fixed_width=40), indel=0.25, max_indel=30)
    _run_case(gpu_ctx, rng, 4, 2000, 4000, dict(band_mode=1, fixed_width=200), indel=0.25, max_indel=60)
    _run_case(gpu_ctx, rng, 3, 1500, 3000, dict(band_mode=1, fixed_width=400), indel=0.25, max_indel=60)


def test_sweeps_that_meet_in_the_middle(gpu_ctx, monkeypatch):
    """k_dp_mid_rs (a read's forward and backward sweeps on two wavefronts that start at the two ends, meet in the middle and go on
    against each other's stored rows; the second halves emit candidates that one pass rescales by the forward sweep's own total):
    what every one-wavefront task of 64+ anti-diagonals runs by default.  Same bits as the mirror of k_dp_rs -- and as k_dp_rs itself
    (NPR_OPT_PAIR = 1), pair list for pair list once both are sorted -- over the three frame classes, bands that drift (rebases,
    also across the cut), odd and even numbers of anti-diagonals, every remainder of D modulo 32 (where the cut and the last block
    fall), reads around the 64-anti-diagonal limit below which a task stays with k_dp_rs, single-base reads."""
    from nanopore_amd import realign as R
    rng = np.random.default_rng(61)
    out = _run_case(gpu_ctx, rng, 24, 1, 400, dict(band_mode=1, fixed_width=40))
    out += _run_case(gpu_ctx, rng, 40, 28, 52, dict(band_mode=1, fixed_width=40))     # D = 56 .. 104: either side of the limit, every D mod 32
    out += _run_case(gpu_ctx, rng, 8, 300, 1500, dict(band_mode=1, fixed_width=200), indel=0.2, max_indel=40)
    out += _run_case(gpu_ctx, rng, 4, 400, 900, dict(band_mode=1, fixed_width=400), indel=0.2, max_indel=30)
    # a read of a handful of bases can miss its range certificate (its start rows) and run again per cell: _run_case compared it
    # with the per-cell mirror then
    assert sum(o["seg_arith"] == [1] for o in out) >= len(out) - 2 and all(len(o["seg_arith"]) == 1 for o in out)
    cases = [random_pair(rng, int(L)) for L in (300, 300, 300, 300, 300, 20, 31, 33)]
    refs, reads, guides = ([bytes(b"ACGT"[c] for c in X) for X, _, _ in cases], [bytes(b"ACGT"[c] for c in Y) for _, Y, _ in cases], [g for _, _, g in cases])
    P = R.make_params(band_mode=1, fixed_width=40)
    b = gpu_ctx.stage(P, refs, reads, guides)
    tasks, _ = b.class_stats()
    b.close()
    short = sum(len(X) + len(Y) < 64 for X, Y, _ in cases)
    assert tasks[12] == len(cases) - short and tasks[15] == short and short >= 2, tasks
    # against k_dp_rs on the device: the same pairs, bit for bit, whatever order the two wavefronts left them in
    mid = gpu_ctx.realign(P, refs, reads, guides, want_pairs=True)
    gpu_ctx.set_option(_lib.OPTIONS["pair"], 1)
    b = gpu_ctx.stage(P, refs, reads, guides)
    tasks, _ = b.class_stats()
    b.close()
    assert tasks[12] == 0 and tasks[15] == len(cases)
    one = gpu_ctx.realign(P, refs, reads, guides, want_pairs=True)
    for u, v in zip(mid, one):
        ku, kv = np.lexsort((u["y"], u["x"])), np.lexsort((v["y"], v["x"]))
        assert np.array_equal(u["x"][ku], v["x"][kv]) and np.array_equal(u["y"][ku], v["y"][kv])
        assert np.array_equal(u["p"][ku].view(np.uint32), v["p"][kv].view(np.uint32))
        assert u["ops"] == v["ops"] and u["loglik"] == v["loglik"] and u["loglik_bwd"] == v["loglik_bwd"] and u["score"] == v["score"]
    # ... and with pair lists that are nearly full: wavefront 1's candidates are moved down over a free gap of a few dozen to a few hundred
    # entries, every block read by both wavefronts before either writes (a gap of 129 .. 255 entries, or any gap below the backward half's
    # length, could lose pairs silently in round 5: the two wavefronts met only for gaps below 128).  Two pairs per base of capacity against
    # the ~1.8 found; a read whose list overflows runs again with four times the capacity, as everywhere.
    gpu_ctx.set_option(_lib.OPTIONS["pair"], 0)
    cases = [random_pair(rng, int(L)) for L in rng.integers(200, 1300, 32)]
    refs, reads, guides = ([bytes(b"ACGT"[c] for c in X) for X, _, _ in cases], [bytes(b"ACGT"[c] for c in Y) for _, Y, _ in cases], [g for _, _, g in cases])
    P2 = R.make_params(band_mode=1, fixed_width=40, max_pairs_per_base=2)
    mid = gpu_ctx.realign(P2, refs, reads, guides, want_pairs=True)
    gpu_ctx.set_option(_lib.OPTIONS["pair"], 1)
    one = gpu_ctx.realign(P2, refs, reads, guides, want_pairs=True)
    gpu_ctx.set_option(_lib.OPTIONS["pair"], 0)
    gaps = [2 * min(len(X), len(Y)) + 64 - len(u["x"]) for (X, Y, _), u in zip(cases, mid)]
    assert sum(0 <= g < 128 for g in gaps) >= 2 and sum(128 <= g < 256 for g in gaps) >= 2, gaps
    for u, v in zip(mid, one):
        ku, kv = np.lexsort((u["y"], u["x"])), np.lexsort((v["y"], v["x"]))
        assert np.array_equal(u["x"][ku], v["x"][kv]) and np.array_equal(u["y"][ku], v["y"][kv])
        assert np.array_equal(u["p"][ku].view(np.uint32), v["p"][kv].view(np.uint32))
        assert u["ops"] == v["ops"] and u["score"] == v["score"]


def test_a_context_closes_the_batches_still_staged_on_it():
    """A batch left open (a caller that raised half way) must not outlive its context: npr_batch_destroy on a batch whose context is gone reads freed
    memory -- at interpreter exit that was a core dump after any failing test.  Context.close() closes what is still staged on it first."""
    from nanopore_amd import realign as R
    ctx = R.Context(0)
    ctx.set_hmm(_hmm_obj("blasr_hmm_0.txt"))
    rng = np.random.default_rng(3)
    X, Y, g = random_pair(rng, 300)
    b = ctx.stage(R.make_params(band_mode=1, fixed_width=60), [bytes(b"ACGT"[c] for c in X)], [bytes(b"ACGT"[c] for c in Y)], [g])
    b.run()
    ctx.close()
    assert b._h is None
    b.close()   # (nothing left to do)
    del b
