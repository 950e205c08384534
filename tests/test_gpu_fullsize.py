"""BASELINE.json-sized inputs on the GPU, checked through size-independent properties (and a sample against the
oracle): forward total == backward total, a base is paired at most once (posterior row / column sums <= 1),
output cigars are global, per-read-type models (config 5) select the right tables."""
import numpy as np
import pytest

from nanopore_amd import _lib

from helpers import cigar_spans, load_model_arrays, orc, seg_arith_of

pytestmark = pytest.mark.gpu


def _codes(buf):
    from nanopore_amd.realign import encode
    return encode(bytes(buf))


def _check_invariants(w, res, off, ops, poff, px, py, pp):
    n = len(w["ref_off"]) - 1
    assert (res["status"] == 0).all()
    assert np.allclose(res["loglik"], res["loglik_bwd"], rtol=2e-6)
    for i in range(n):
        o = ops[off[i]:off[i + 1]]
        lx, ly = w["ref_off"][i + 1] - w["ref_off"][i], w["read_off"][i + 1] - w["read_off"][i]
        assert int(o[o[:, 0] != 1, 1].sum()) == lx and int(o[o[:, 0] != 2, 1].sum()) == ly
        x, y, p = px[poff[i]:poff[i + 1]], py[poff[i]:poff[i + 1]], pp[poff[i]:poff[i + 1]]
        assert (p >= 0.01).all() and p.max() <= 1.0 + 1e-5
        assert np.bincount(y, weights=p, minlength=ly).max() <= 1.0 + 1e-4
        assert np.bincount(x, weights=p, minlength=lx).max() <= 1.0 + 1e-4
        assert (np.diff(x.astype(np.int64) * (ly + 1) + y) > 0).all()          # sorted by (x, y), no duplicates


def test_north_star_shape_properties_and_oracle_sample(gpu_ctx):
    """~10 kb reads x 50 kb slices, W = 200 (the shape the north-star target is quoted on)."""
    from nanopore_amd import realign as R, synth
    from nanopore_amd.hmm import Hmm
    from helpers import MODEL_DIR
    T, E, _ = load_model_arrays()
    w, W = synth.config_north_star(T, E, n_reads=96, windowed=False)  # flanks spelled out as 20 kb deletions
    gpu_ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + "/blasr_hmm_0.txt"))
    b = gpu_ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w["ref"], w["ref_off"], w["read"],
                          w["read_off"], w["guide_ops"], w["guide_off"])
    assert b.stats()["kernel_variant"] == 1
    b.run()
    b.finish()
    res = b.results()
    off, ops = b.ops()
    poff, px, py, pp = b.pairs()
    arith = seg_arith_of(b)
    b.close()
    _check_invariants(w, res, off, ops, poff, px, py, pp)
    assert res["cells"].min() > 1e6
    # the realigner recovers from the degraded guide: most reads come back closer to the truth than the guide
    # sample: two reads against the oracle's fp32 mirror (bit-exact) -- ~4e6 cells each
    h = orc.make_hmm(T, E)
    P = orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=W)
    for i in (0, 57):
        X = _codes(w["ref"][w["ref_off"][i]:w["ref_off"][i + 1]])
        Y = _codes(w["read"][w["read_off"][i]:w["read_off"][i + 1]])
        g = [tuple(int(v) for v in r) for r in w["guide_ops"][w["guide_off"][i]:w["guide_off"][i + 1]]]
        m = orc.realign_read(h, P, X, Y, g, precision=1, seg_arith=arith(i))
        assert m["cells"] == res["cells"][i]
        assert [tuple(int(v) for v in r) for r in ops[off[i]:off[i + 1]]] == m["ops"]
        order = np.lexsort((m["py"], m["px"]))
        assert np.array_equal(pp[poff[i]:poff[i + 1]], m["pp"].astype(np.float32)[order])
        # ... and against the fp64 log-space oracle at this shape: every posterior within the 1e-4 north_star states (a pair may sit on
        # either side of the 0.01 threshold), the cigar identical or differing only where an exact tie is placed, the likelihood to 2e-6
        m64 = orc.realign_read(h, P, X, Y, g, precision=0)
        dev = {(int(a), int(c)): float(v) for a, c, v in zip(px[poff[i]:poff[i + 1]], py[poff[i]:poff[i + 1]], pp[poff[i]:poff[i + 1]])}
        ref = {(int(a), int(c)): float(v) for a, c, v in zip(m64["px"], m64["py"], m64["pp"])}
        worst = 0.0
        for k in set(dev) | set(ref):
            a, c = dev.get(k), ref.get(k)
            if a is None or c is None:
                assert abs((a if a is not None else c) - 0.01) < 1e-4
            else:
                worst = max(worst, abs(a - c))
        assert worst < 1e-4, worst
        assert res["loglik"][i] == pytest.approx(m64["total_ll"], rel=2e-6)
        assert cigar_spans(m64["ops"]) == cigar_spans([tuple(int(v) for v in r) for r in ops[off[i]:off[i + 1]]])


def test_north_star_windowed_guides_match_explicit_slices(gpu_ctx):
    """The bench's default shape: the guide carries the coordinates of its window inside the 50 kb slice
    (npr_batch_create_at).  Same results as staging the windows cut out by hand; posterior coordinates absolute."""
    from nanopore_amd import realign as R, synth
    from nanopore_amd.hmm import Hmm
    from helpers import MODEL_DIR
    T, E, _ = load_model_arrays()
    w, W = synth.config_north_star(T, E, n_reads=64)
    gpu_ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + "/blasr_hmm_0.txt"))
    P = R.make_params(band_mode=R.BAND_FIXED, fixed_width=W)
    b = gpu_ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"],
                          guide_start=w["guide_start"])
    assert b.stats()["kernel_variant"] == 1
    b.run(), b.finish()
    res, (off, ops), (poff, px, py, pp) = b.results(), b.ops(), b.pairs()
    arith = seg_arith_of(b)
    b.close()
    # the same windows cut out on the host
    lead, ilen = w["lead"], w["interval_len"]
    cut = np.concatenate([w["ref"][w["ref_off"][i] + lead[i]:w["ref_off"][i] + lead[i] + ilen[i]] for i in range(64)])
    cut_off = np.concatenate([[0], np.cumsum(ilen)])
    c = gpu_ctx.stage_csr(P, cut, cut_off, w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
    c.run(), c.finish()
    res2, (off2, ops2), (poff2, qx, qy, qp) = c.results(), c.ops(), c.pairs()
    c.close()
    assert (res["status"] == 0).all() and np.array_equal(res["cells"], res2["cells"]) and res["cells"].max() < 3e6
    assert np.array_equal(off, off2) and np.array_equal(ops, ops2) and np.array_equal(res["score"], res2["score"])
    assert np.array_equal(poff, poff2) and np.array_equal(pp, qp) and np.array_equal(py, qy)
    assert np.array_equal(px, qx + np.repeat(lead, np.diff(poff)).astype(np.int32))
    # a window that sticks out of the slice is refused for that read only
    bad = w["guide_start"].copy()
    bad[3, 0] = 50000 - ilen[3] + 1
    d = gpu_ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], guide_start=bad)
    d.run(), d.finish()
    st = d.results()["status"]
    d.close()
    assert st[3] != 0 and (np.delete(st, 3) == 0).all()
    # one read against the oracle's fp32 mirror on the cut-out window
    h = orc.make_hmm(T, E)
    PO = orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=W)
    i = 11
    X = _codes(cut[cut_off[i]:cut_off[i + 1]])
    Y = _codes(w["read"][w["read_off"][i]:w["read_off"][i + 1]])
    g = [tuple(int(v) for v in r) for r in w["guide_ops"][w["guide_off"][i]:w["guide_off"][i + 1]]]
    m = orc.realign_read(h, PO, X, Y, g, precision=1, seg_arith=arith(i))
    assert m["cells"] == res["cells"][i]
    assert [tuple(int(v) for v in r) for r in ops[off[i]:off[i + 1]]] == m["ops"]
    order = np.lexsort((m["py"], m["px"]))
    assert np.array_equal(pp[poff[i]:poff[i + 1]], m["pp"].astype(np.float32)[order])
    assert np.array_equal(px[poff[i]:poff[i + 1]], m["px"][order].astype(np.int32) + lead[i])


def test_reference_anchor_band_wide_register_kernel(gpu_ctx, monkeypatch):
    """The reference's own band (anchors +- diagonalExpansion 10, trim 14, splitMatrixBiggerThanThis 3000,
    nanopore/analyses/utils.py:587) on reads from the shipped nanopore model: few anchors survive the trimming, the
    bands are diamonds hundreds to thousands of cells wide, and the multi-wavefront register kernel (k_dp_wide) takes
    them.  Its results must be the generic kernel's bit for bit, and the fp32 mirror's."""
    from nanopore_amd import realign as R, synth
    from nanopore_amd.hmm import Hmm
    from helpers import MODEL_DIR
    T, E, _ = load_model_arrays()
    w = synth.make_workload(1007, 48, 3000, T, E, flank=0, length_sigma=0.5, len_min=300, len_max=9000)
    gpu_ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + "/blasr_hmm_0.txt"))
    P = R.make_params(band_mode=R.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=3000)
    gpu_ctx.set_option(_lib.OPTIONS["no_tile"], 1)  # the stripe kernel (tests/test_gpu_tile.py) would take these bands

    def run():
        b = gpu_ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
        tasks, cells = b.class_stats()
        b.run(), b.finish()
        out = (b.results(), b.ops(), b.pairs(), tasks, cells, b.stats()["max_width"], seg_arith_of(b))
        b.close()
        return out

    # ragged segment ends inside the wide kernel: a small split threshold cuts the big rectangles, both halves keep
    # up to 700 cells of them and start / end in the long-gap states
    P_keep = P
    P = R.make_params(band_mode=R.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=700,
                      max_pairs_per_base=40)  # ragged ends blur the posteriors: more pairs per base than the default 6
    r1, o1, p1, t1, _, w1, _ = run()
    gpu_ctx.set_option(_lib.OPTIONS["no_wide"], 1)
    r2, o2, p2, t2, _, _, _ = run()
    gpu_ctx.set_option(_lib.OPTIONS["no_wide"], 0)
    assert (r1["status"] == 0).all() and r1["n_segments"].max() > 1 and t1[3:7].sum() > 0 and t2[3:7].sum() == 0 and 256 <= w1 <= 1400
    assert np.array_equal(r1["loglik"], r2["loglik"]) and np.array_equal(r1["loglik_bwd"], r2["loglik_bwd"])
    assert np.array_equal(o1[1], o2[1]) and np.array_equal(p1[3], p2[3]) and np.array_equal(p1[1], p2[1])
    P = P_keep

    res, (off, ops), (poff, px, py, pp), tasks, cells, maxw, arith = run()
    assert (res["status"] == 0).all() and maxw > 1024
    assert tasks[3:7].sum() > 0.5 * tasks.sum() and cells[3:7].sum() > 0.9 * cells.sum()   # k_dp_wide did the work
    assert (tasks[3:7] > 0).sum() >= 3                                                      # in several frame sizes
    gpu_ctx.set_option(_lib.OPTIONS["no_wide"], 1)
    res2, (off2, ops2), (poff2, qx, qy, qp), tasks2, _, _, _ = run()
    assert tasks2[3:7].sum() == 0 and tasks2[7:].sum() == tasks[3:].sum()
    assert np.array_equal(res["cells"], res2["cells"]) and np.array_equal(res["loglik"], res2["loglik"])
    assert np.array_equal(res["loglik_bwd"], res2["loglik_bwd"]) and np.array_equal(res["score"], res2["score"])
    assert np.array_equal(off, off2) and np.array_equal(ops, ops2)
    assert np.array_equal(poff, poff2) and np.array_equal(px, qx) and np.array_equal(py, qy) and np.array_equal(pp, qp)
    assert np.abs(res["loglik"] - res["loglik_bwd"]).max() < 1e-2
    # two reads against the oracle's fp32 mirror
    h = orc.make_hmm(T, E)
    PO = orc.make_params(band_mode=orc.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=3000)
    order_by_cells = np.argsort(res["cells"])
    for i in (int(order_by_cells[len(order_by_cells) // 2]), int(order_by_cells[5])):
        X = _codes(w["ref"][w["ref_off"][i]:w["ref_off"][i + 1]])
        Y = _codes(w["read"][w["read_off"][i]:w["read_off"][i + 1]])
        g = [tuple(int(v) for v in r) for r in w["guide_ops"][w["guide_off"][i]:w["guide_off"][i + 1]]]
        m = orc.realign_read(h, PO, X, Y, g, precision=1, seg_arith=arith(i))
        assert m["cells"] == res["cells"][i]
        assert [tuple(int(v) for v in r) for r in ops[off[i]:off[i + 1]]] == m["ops"]
        order = np.lexsort((m["py"], m["px"]))
        assert np.array_equal(pp[poff[i]:poff[i + 1]], m["pp"].astype(np.float32)[order])


def test_long_reads_and_per_read_type_models(gpu_ctx):
    """Config 5 in miniature: 10-50 kb reads, three read types with their own HMM slot (hmm_0 / 20 / 40)."""
    from nanopore_amd import realign as R, synth
    from nanopore_amd.hmm import Hmm
    from helpers import MODEL_DIR
    T, E, _ = load_model_arrays()
    w = synth.make_workload(1005, 12, 30000, T, E, flank=400, uniform_len=(10000, 50000))
    slot = np.arange(12, dtype=np.int32) % 3
    for s, name in enumerate(("blasr_hmm_0.txt", "blasr_hmm_20.txt", "blasr_hmm_40.txt")):
        gpu_ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + "/" + name), slot=s)
    P = R.make_params(band_mode=R.BAND_FIXED, fixed_width=200)
    b = gpu_ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], model_slot=slot)
    b.run()
    b.finish()
    res = b.results()
    off, ops = b.ops()
    poff, px, py, pp = b.pairs()
    arith = seg_arith_of(b)
    b.close()
    _check_invariants(w, res, off, ops, poff, px, py, pp)
    assert (w["read_off"][1:] - w["read_off"][:-1]).max() > 30000
    # each read used ITS model: the oracle with the matching model reproduces the log-likelihood, another does not
    PO = orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=200)
    for i in (0, 1, 2):
        X = _codes(w["ref"][w["ref_off"][i]:w["ref_off"][i + 1]])
        Y = _codes(w["read"][w["read_off"][i]:w["read_off"][i + 1]])
        g = [tuple(int(v) for v in r) for r in w["guide_ops"][w["guide_off"][i]:w["guide_off"][i + 1]]]
        lls = []
        for name in ("blasr_hmm_0.txt", "blasr_hmm_20.txt", "blasr_hmm_40.txt"):
            Tm, Em, _ = load_model_arrays(name)
            lls.append(orc.realign_read(orc.make_hmm(Tm, Em), PO, X, Y, g, precision=1, seg_arith=arith(i))["total_ll"])
        assert res["loglik"][i] == pytest.approx(lls[slot[i]], rel=1e-12)
        assert all(abs(res["loglik"][i] - lls[k]) > 1.0 for k in range(3) if k != slot[i])
    gpu_ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + "/blasr_hmm_0.txt"), slot=0)


def test_invalid_model_slot_is_a_per_read_error(gpu_ctx):
    from nanopore_amd import realign as R
    P = R.make_params(band_mode=R.BAND_FIXED, fixed_width=20)
    out = gpu_ctx.realign(P, [b"ACGTACGT", b"ACGTACGT"], [b"ACGTACGT", b"ACGTACGT"], [[(0, 8)], [(0, 8)]], model_slot=[0, 7])
    assert out[0]["status"] == 0 and out[1]["status"] == -4
