"""BASELINE.json's configurations on the GPU, each with its own generator and at its named size:
  configs[1]  1 000 reads x 1 kb, band 100: EVERY read against the oracle (cigars + scores vs the fp32 mirror and the
              fp64 log-space oracle);
  configs[2]  50 000 reads ~8 kb on ONE shared 4.6 Mb contig (ref_index, guides with window coordinates), band 200:
              size-independent properties over all reads, a 32-read sample bit-exact vs the mirror / 1e-4 vs fp64;
  configs[3]  the same kind of set sharded over TWO ranks (one GPU shared, gloo collectives): contiguous shards, the real
              stage / run / finish on each rank, each rank's block of the SAM written in place, scalars gathered for the
              XML -- byte-identical to the one-rank job;
  configs[4]  3 x 1 000 reads of 10-50 kb with per-read-type model slots (hmm_0 / hmm_20 / hmm_40).
PARITY UNPINNED: the oracle is this build's restatement of cactus_realign (absent from the reference snapshot)."""
import os
import socket

import numpy as np
import pytest

from helpers import MODEL_DIR, load_model_arrays, orc, seg_arith_of

pytestmark = pytest.mark.gpu


def _codes(buf):
    from nanopore_amd.realign import encode
    return encode(bytes(buf))


def _guide(w, i):
    return [tuple(int(v) for v in r) for r in w["guide_ops"][w["guide_off"][i]:w["guide_off"][i + 1]]]


def _window(w, i):
    """(reference codes of the guide's window, read codes) of read i, whatever form the workload has."""
    k = int(w["ref_index"][i]) if w.get("ref_index") is not None else i
    lo = int(w["ref_off"][k])
    if w.get("guide_start") is not None:
        lo += int(w["guide_start"][i][0])
        hi = lo + int(w["interval_len"][i])
    else:
        hi = int(w["ref_off"][k + 1])
    return _codes(w["ref"][lo:hi]), _codes(w["read"][w["read_off"][i]:w["read_off"][i + 1]])


def _set_models(ctx, names=("blasr_hmm_0.txt",)):
    from nanopore_amd.hmm import Hmm
    for s, name in enumerate(names):
        ctx.set_hmm(Hmm.loadHmm(os.path.join(MODEL_DIR, name)), slot=s)


def _cigar_spans(ops, off, i):
    o = ops[off[i]:off[i + 1]]
    return int(o[o[:, 0] != 1, 1].sum()), int(o[o[:, 0] != 2, 1].sum())


def test_config2_every_read_against_the_oracle(gpu_ctx):
    from nanopore_amd import realign as R, synth
    T, E, _ = load_model_arrays()
    w, W = synth.config_c2(T, E, n_reads=1000)
    _set_models(gpu_ctx)
    b = gpu_ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w["ref"], w["ref_off"], w["read"], w["read_off"],
                          w["guide_ops"], w["guide_off"])
    b.run(), b.finish()
    res, (off, ops) = b.results(), b.ops()
    seg_arith = b.segment_arith()  # the mirror restates the arithmetic of the kernel class each segment ran in
    b.close()
    assert (res["status"] == 0).all() and len(res) == 1000
    h = orc.make_hmm(T, E)
    P = orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=W)
    X, Y = _codes(w["ref"]), _codes(w["read"])
    m32 = orc.realign_batch(h, P, X, w["ref_off"], Y, w["read_off"], w["guide_ops"], w["guide_off"], precision=1, threads=8, native=True,
                            seg_arith=seg_arith)
    m64 = orc.realign_batch(h, P, X, w["ref_off"], Y, w["read_off"], w["guide_ops"], w["guide_off"], precision=0, threads=8, native=True)
    assert np.array_equal(res["cells"], m32["cells"]) and int(res["cells"].sum()) > 9e7
    same64 = 0
    for i in range(1000):
        g = ops[off[i]:off[i + 1]]
        assert np.array_equal(g, m32["ops"][i]), "read %d: cigar differs from the fp32 mirror" % i
        same64 += int(np.array_equal(g, m64["ops"][i]))
    assert np.allclose(res["score"], m32["score"], rtol=0, atol=1e-12)
    assert np.allclose(res["loglik"], m32["total_ll"], rtol=1e-12, atol=1e-9)
    assert np.allclose(res["loglik"], m64["total_ll"], rtol=2e-6)
    assert np.abs(res["score"] - m64["score"]).max() < 1e-4
    # fp32 (device = mirror, bit for bit) against the fp64 oracle: exact-tie placements may differ.  The
    # count is printed (pytest -s / the captured output of a failure); for scale: the reference's own piecewise log-add, if
    # recalled right, moves 262 of these 1000 cigars against exact arithmetic (tools/logadd_risk.py)
    ties = [i for i in range(1000) if not np.array_equal(ops[off[i]:off[i + 1]], m64["ops"][i])]
    print("configs[1]: %d/1000 cigars identical to the fp64 oracle's; differing reads: %s" % (same64, ties))
    for i in ties:  # a tie moves an indel along a repeat: same spans, same number of aligned pairs +- a handful, scores within 1e-4
        a, b_ = ops[off[i]:off[i + 1]], m64["ops"][i]
        assert abs(int(a[a[:, 0] == 0, 1].sum()) - int(b_[b_[:, 0] == 0, 1].sum())) <= 8
    # north_star asks for bit-exact cigars: on this configuration -- seeded, deterministic on the device -- every one of the 1000 is also the fp64
    # oracle's (rounds 3-5 accepted 995; the count was 1000 in every run).  A change of the workload generator that brings a tie in will fail here
    # and be looked at, not absorbed.
    assert same64 == 1000


def test_config3_shared_contig_50k_reads(gpu_ctx):
    from nanopore_amd import realign as R, synth
    T, E, _ = load_model_arrays()
    w, W = synth.config_c3_shared(T, E, n_reads=50000)
    n = 50000
    _set_models(gpu_ctx)
    P = R.make_params(band_mode=R.BAND_FIXED, fixed_width=W)
    b = gpu_ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"],
                          ref_index=w["ref_index"], guide_start=w["guide_start"])
    st = b.stats()
    assert st["n_reads"] == n and st["kernel_variant"] == 1 and st["cells"] > 7e10
    b.run(), b.finish()
    res, (off, ops) = b.results(), b.ops()
    b.close()
    # properties over all 50 k reads
    assert (res["status"] == 0).all()
    assert np.allclose(res["loglik"], res["loglik_bwd"], rtol=2e-6)
    assert (res["score"] > 0.3).all() and (res["score"] <= 1.0).all()
    ilen = w["interval_len"]
    rlen = w["read_off"][1:] - w["read_off"][:-1]
    isM, isI, isD = (ops[:, 0] == k for k in (0, 1, 2))
    csum = lambda m: np.concatenate([[0], np.cumsum(np.where(m, ops[:, 1], 0))])  # noqa: E731
    cm, ci, cd = csum(isM), csum(isI), csum(isD)
    span = lambda c: c[off[1:]] - c[off[:-1]]  # noqa: E731
    assert np.array_equal(span(cm) + span(cd), ilen) and np.array_equal(span(cm) + span(ci), rlen)  # cigars global over the windows
    assert (ops[:, 1] > 0).all() and (ops[:, 0] <= 2).all()
    same_op_next = ops[1:, 0] == ops[:-1, 0]
    same_op_next[off[1:-1] - 1] = False
    assert not same_op_next.any()                                                                # run-length merged
    # pairs and window coordinates on a re-staged sample of 512 reads; 32 of them against the oracle
    idx = np.arange(0, n, n // 512)[:512]
    sub = synth.take_reads(w, idx)
    c = gpu_ctx.stage_csr(P, sub["ref"], sub["ref_off"], sub["read"], sub["read_off"], sub["guide_ops"], sub["guide_off"],
                          ref_index=sub["ref_index"], guide_start=sub["guide_start"])
    c.run(), c.finish()
    res2, (off2, ops2), (poff, px, py, pp) = c.results(), c.ops(), c.pairs()
    arith2 = seg_arith_of(c)
    c.close()
    assert np.array_equal(res2["loglik"], res["loglik"][idx]) and np.array_equal(res2["score"], res["score"][idx])
    h = orc.make_hmm(T, E)
    PO = orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=W)
    for k in range(512):
        i = int(idx[k])
        assert np.array_equal(ops2[off2[k]:off2[k + 1]], ops[off[i]:off[i + 1]])
        x, y, p = px[poff[k]:poff[k + 1]], py[poff[k]:poff[k + 1]], pp[poff[k]:poff[k + 1]]
        x0 = int(w["guide_start"][i][0])
        assert x.min() >= x0 and x.max() < x0 + ilen[i] and y.min() >= 0 and y.max() < rlen[i]   # absolute contig coordinates
        assert (p >= 0.01).all() and np.bincount(y, weights=p).max() <= 1.0 + 1e-4
        assert np.bincount(x - x0, weights=p).max() <= 1.0 + 1e-4
        if k % 16 == 0:
            X, Y = _window(w, i)
            g = _guide(w, i)
            m32 = orc.realign_read(h, PO, X, Y, g, precision=1, seg_arith=arith2(k))
            m64 = orc.realign_read(h, PO, X, Y, g, precision=0)
            assert m32["cells"] == res["cells"][i]
            assert [tuple(int(v) for v in r) for r in ops[off[i]:off[i + 1]]] == m32["ops"]
            order = np.lexsort((m32["py"], m32["px"]))
            assert np.array_equal(p, m32["pp"].astype(np.float32)[order])
            assert np.array_equal(x - x0, m32["px"][order].astype(np.int32))
            d64 = {(int(a), int(c_)): float(v) for a, c_, v in zip(m64["px"], m64["py"], m64["pp"])}
            for a, c_, v in zip(x - x0, y, p):
                assert abs(d64.get((int(a), int(c_)), 0.01) - float(v)) < 1e-4
            assert res["loglik"][i] == pytest.approx(m64["total_ll"], rel=2e-6)


def test_config5_three_read_types_with_their_own_models(gpu_ctx):
    from nanopore_amd import realign as R, synth
    T, E, _ = load_model_arrays()
    names = ("blasr_hmm_0.txt", "blasr_hmm_20.txt", "blasr_hmm_40.txt")
    w, W, slot = synth.config_c5(T, E, n_reads_per_type=1000)
    n = 3000
    _set_models(gpu_ctx, names)
    P = R.make_params(band_mode=R.BAND_FIXED, fixed_width=W)
    b = gpu_ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], model_slot=slot)
    b.run(), b.finish()
    res, (off, ops) = b.results(), b.ops()
    arith = seg_arith_of(b)
    b.close()
    _set_models(gpu_ctx)
    rlen = w["read_off"][1:] - w["read_off"][:-1]
    xlen = w["ref_off"][1:] - w["ref_off"][:-1]
    assert (res["status"] == 0).all() and rlen.max() > 45000 and rlen.min() < 12000 and res["cells"].sum() > 1.5e10
    assert np.allclose(res["loglik"], res["loglik_bwd"], rtol=2e-6)
    for i in range(0, n, 7):
        assert _cigar_spans(ops, off, i) == (int(xlen[i]), int(rlen[i]))
    # every slot used ITS model: the mirror with the matching model reproduces cigar and likelihood bit for bit,
    # the other two models give another likelihood
    PO = orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=W)
    hm = [orc.make_hmm(*load_model_arrays(nm)[:2]) for nm in names]
    shortest = [int(np.flatnonzero(slot == s)[np.argsort(rlen[slot == s])[k]]) for s in range(3) for k in range(4)]
    for i in shortest:
        X, Y = _window(w, i)
        g = _guide(w, i)
        lls = [orc.realign_read(hm[s], PO, X, Y, g, precision=1, seg_arith=arith(i)) for s in range(3)]
        m = lls[int(slot[i])]
        assert res["loglik"][i] == pytest.approx(m["total_ll"], rel=1e-12)
        assert [tuple(int(v) for v in r) for r in ops[off[i]:off[i + 1]]] == m["ops"]
        assert all(abs(res["loglik"][i] - lls[s]["total_ll"]) > 1.0 for s in range(3) if s != slot[i])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, n_reads, out_dir):
    """One rank of the sharded job: its own context on cuda:0 (the box has one GPU; RCCL refuses two ranks on one device,
    so the collectives go over gloo -- the code path is the one bench.py --workload c3 runs under RCCL)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["NPR_HOST_THREADS"] = "4"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nanopore_amd import dist as npd, job, realign as R, synth
    from nanopore_amd.hmm import Hmm
    T, E, _ = load_model_arrays()
    w, W = synth.config_c3_shared(T, E, n_reads=n_reads, genome_len=400000)
    ctx = R.Context(0)
    ctx.set_hmm(Hmm.loadHmm(os.path.join(MODEL_DIR, "blasr_hmm_0.txt")))
    out = job.run_job(ctx, R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w, out_dir=out_dir, device="cpu")
    if rank == 0:
        np.savez(os.path.join(out_dir, "merged.npz"), status=out["status"], score=out["score"], n_ops=out["n_ops"])
    else:
        assert set(out) == {"timings"}
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_shard_one_read_set_and_match_the_single_rank_job(gpu_ctx, tmp_path):
    import torch.multiprocessing as mp
    from nanopore_amd import dist as npd, job, realign as R, synth
    n_reads = 1536
    T, E, _ = load_model_arrays()
    w, W = synth.config_c3_shared(T, E, n_reads=n_reads, genome_len=400000)
    _set_models(gpu_ctx)
    one = job.run_job(gpu_ctx, R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w, out_dir=str(tmp_path / "one"))
    assert (one["status"] == 0).all() and one["timings"]["cells"] > 1e9
    # the shards are contiguous, partition the set and balance the work
    work = w["read_off"][1:] - w["read_off"][:-1]
    bounds = npd.shard_ranges(work, 2)
    assert bounds[0] == 0 and bounds[2] == n_reads and 0 < bounds[1] < n_reads
    assert abs(int(work[:bounds[1]].sum()) - int(work[bounds[1]:].sum())) < 0.02 * work.sum()
    two_dir = str(tmp_path / "two")
    os.makedirs(two_dir)
    mp.spawn(_rank_main, args=(2, _free_port(), n_reads, two_dir), nprocs=2, join=True)
    z = np.load(os.path.join(two_dir, "merged.npz"))
    for k in ("status", "score", "n_ops"):
        assert np.array_equal(z[k], one[k]), k
    sam1 = open(one["sam"], "rb").read()
    sam2 = open(os.path.join(two_dir, "realigned.sam"), "rb").read()
    assert sam1 == sam2 and sam1.count(b"\n") == n_reads + 2
    assert open(one["xml"], "rb").read() == open(os.path.join(two_dir, "summary.xml"), "rb").read()
    # the SAM is the input order with the realigner's cigars: spot-check records against a direct realignment
    b = gpu_ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"],
                          w["guide_off"], ref_index=w["ref_index"], guide_start=w["guide_start"])
    b.run(), b.finish()
    off, ops = b.ops()
    b.close()
    lines = sam1.split(b"\n")[2:]
    for i in (0, 1, int(bounds[1]) - 1, int(bounds[1]), n_reads - 1):
        f = lines[i].split(b"\t")
        assert f[0] == b"read_%d" % i and int(f[3]) == int(w["guide_start"][i][0]) + 1
        cig = b"".join(b"%d%s" % (ln, b"MID"[op:op + 1]) for op, ln in ops[off[i]:off[i + 1]])
        assert f[5] == cig and f[9] == bytes(w["read"][w["read_off"][i]:w["read_off"][i + 1]])
        assert one["n_ops"][i] == off[i + 1] - off[i]
