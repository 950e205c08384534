"""The files -> files job behind the plugin surface (nanopore_amd/job.py; analyses.utils.realignSamFile,
AbstractMapper.realignSamFile): bulk native ingest, records sharded over ranks, two batches in flight per rank, every rank
writing its block of the output -- against the record-at-a-time host mirror (realignSamFileByRecord: Samfile iterator,
one AlignedRead per record, one write per record), byte for byte; on one rank and on two (gloo collectives, one GPU shared:
the code path RCCL runs under backend nccl); BASELINE.json configs[4] at SURVEY's size (3 x 10 000 reads of 10-50 kb, one
model per read type) with the coverage / substitutions XML built from the table the job reduced on the device.
Reference: nanopore/analyses/utils.py:557-609, nanopore/mappers/abstractMapper.py:25-39, nanopore/pipeline.py:114-129."""
import os
import socket
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from helpers import MODEL_DIR, load_model_arrays

pytestmark = pytest.mark.gpu

HMM0 = os.path.join(MODEL_DIR, "blasr_hmm_0.txt")


def _c3_files(tmp, n_reads, windowed=True, seed_genome=400000):
    """A C3-shaped read set as files: reads of ~8 kb (lognormal) cut from one contig.  windowed: ONE shared contig, every
    record local to its window (POS = window start); else per-read reference slices with 400-base flanks and GLOBAL records
    (pos 0, leading / trailing deletions: the shape chainSamFile emits, utils.py:381-382)."""
    from nanopore_amd import synth
    T, E, _ = load_model_arrays()
    if windowed:
        w, _ = synth.config_c3_shared(T, E, n_reads=n_reads, genome_len=seed_genome)
    else:
        rng = np.random.default_rng(5)
        genome = synth.random_reference(rng, seed_genome, 0.5)
        w = synth.make_workload(1007, n_reads, 3000, T, E, flank=400, genome=genome, length_sigma=0.3, len_min=500, len_max=8000)
    sam, fa, fq = (os.path.join(tmp, k) for k in ("in.sam", "ref.fa", "reads.fq"))
    synth.write_workload_files(w, sam, fa, fastq_path=fq)
    return w, sam, fa, fq


@pytest.mark.parametrize("windowed", [True, False], ids=["shared_contig_local_records", "per_read_slices_global_records"])
def test_bulk_job_equals_the_record_path_byte_for_byte(tmp_path, gpu_ctx, monkeypatch, windowed):
    from nanopore_amd import job
    from nanopore_amd.analyses import utils
    n = 768 if windowed else 400
    w, sam, fa, fq = _c3_files(str(tmp_path), n, windowed)
    ref_out, out = str(tmp_path / "by_record.sam"), str(tmp_path / "bulk.sam")
    want = utils.realignSamFileByRecord(sam, ref_out, fq, fa, HMM0, 0.5, 0.0, ctx=gpu_ctx)
    monkeypatch.setattr(job, "CHUNK_BASES", 700000)             # ~8 chunks: both workers busy, blocks written out of step
    res = utils.realignSamFile(sam, out, fq, fa, HMM0, 0.5, 0.0, ctx=gpu_ctx)
    a, b = open(ref_out, "rb").read(), open(out, "rb").read()
    assert a == b and a.count(b"\n") == n + len(w["ref_off"]) + 0   # @HD + one @SQ per reference + n records
    assert len(res) == n and (res["status"] == 0).all()
    assert np.array_equal(res["score"], np.array([r["score"] for r in want]))
    assert np.array_equal(res["loglik"], np.array([r["loglik"] for r in want]))
    assert np.array_equal(res["cells"], np.array([r["cells"] for r in want]))
    # one chunk, one worker: the same bytes again
    monkeypatch.setattr(job, "CHUNK_BASES", 1 << 40)
    monkeypatch.setattr(job, "WORKERS", 1)
    utils.realignSamFile(sam, out, fq, fa, HMM0, 0.5, 0.0, ctx=gpu_ctx)
    assert open(out, "rb").read() == a


def test_records_without_a_reference_are_dropped_and_failures_raise(tmp_path, gpu_ctx):
    from nanopore_amd.analyses import utils
    w, sam, fa, fq = _c3_files(str(tmp_path), 24, True, seed_genome=60000)
    lines = open(sam).read().split("\n")
    head = [l for l in lines if l.startswith("@")]
    recs = [l for l in lines if l and not l.startswith("@")]
    recs.insert(3, "lost\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\t*")        # samIterator drops it (utils.py:287-293)
    mixed = str(tmp_path / "mixed.sam")
    open(mixed, "w").write("\n".join(head + recs) + "\n")
    out, ref_out = str(tmp_path / "o.sam"), str(tmp_path / "r.sam")
    res = utils.realignSamFile(mixed, out, fq, fa, HMM0, 0.5, 0.0, ctx=gpu_ctx)
    utils.realignSamFileByRecord(mixed, ref_out, fq, fa, HMM0, 0.5, 0.0, ctx=gpu_ctx)
    assert len(res) == 24 and open(out, "rb").read() == open(ref_out, "rb").read() and b"lost" not in open(out, "rb").read()
    # a cigar operation outside M I D S H: the reference asserts (utils.py:171)
    f = recs[0].split("\t")
    f[5] = "10M5N10M"
    open(mixed, "w").write("\n".join(head + ["\t".join(f)]) + "\n")
    with pytest.raises(AssertionError):
        utils.realignSamFile(mixed, out, fq, fa, HMM0, 0.5, 0.0, ctx=gpu_ctx)
    # a guide that runs past its reference: the record fails, the job raises and leaves no output (pipeline.py:209-210)
    f = recs[1].split("\t")
    f[3] = str(60000 - 100)
    open(mixed, "w").write("\n".join(head + recs[:1] + ["\t".join(f)]) + "\n")
    with pytest.raises(RuntimeError):
        utils.realignSamFile(mixed, out, fq, fa, HMM0, 0.5, 0.0, ctx=gpu_ctx)
    assert not os.path.exists(out)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, sam, fa, fq, out):
    """One rank of the sharded files -> file job, through the plugin surface (the box has one GPU; RCCL refuses two ranks on
    one device, so the collectives go over gloo)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["NPR_HOST_THREADS"] = "4"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nanopore_amd import job
    from nanopore_amd.analyses import utils
    job.CHUNK_BASES = 900000
    res = utils.realignSamFile(sam, out, fq, fa, HMM0, 0.5, 0.0)
    if rank == 0:
        np.save(out + ".score.npy", res["score"])
    else:
        assert res is None
    dist.barrier()
    job.close_contexts()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_through_the_plugin_surface_match_one_rank(tmp_path, gpu_ctx):
    import torch.multiprocessing as mp
    from nanopore_amd.analyses import utils
    n = 600
    w, sam, fa, fq = _c3_files(str(tmp_path), n, True)
    one = str(tmp_path / "one.sam")
    res = utils.realignSamFileByRecord(sam, one, fq, fa, HMM0, 0.5, 0.0, ctx=gpu_ctx)
    two = str(tmp_path / "two.sam")
    mp.spawn(_rank_main, args=(2, _free_port(), sam, fa, fq, two), nprocs=2, join=True)
    assert open(one, "rb").read() == open(two, "rb").read()
    assert np.array_equal(np.load(two + ".score.npy"), np.array([r["score"] for r in res]))


def _c5_part(args):
    """One slice of the C5 read set (a worker process: the generator walks the error channel base by base)."""
    k, n_per_type, tmp = args
    from nanopore_amd import synth
    T, E, _ = load_model_arrays()
    w = synth.make_workload(1005 + 31 * k, 3 * n_per_type, 30000, T, E, flank=400, uniform_len=(10000, 50000), windowed=True)
    paths = []
    for t in range(3):
        sub = synth.take_reads(w, np.arange(t * n_per_type, (t + 1) * n_per_type))
        sub["guide_start"] = np.ascontiguousarray(sub["guide_start"])
        p = os.path.join(tmp, "part_%d_%d.npz" % (t, k))
        np.savez(p, **{key: sub[key] for key in ("ref", "ref_off", "read", "read_off", "guide_ops", "guide_off", "guide_start")})
        paths.append(p)
    return paths


def _concat(parts):
    out = {}
    for key in ("ref", "read", "guide_ops", "guide_start"):
        out[key] = np.concatenate([p[key] for p in parts])
    for key, unit in (("ref_off", "ref"), ("read_off", "read"), ("guide_off", "guide_ops")):
        offs, base = [np.zeros(1, dtype=np.int64)], 0
        for p in parts:
            offs.append(p[key][1:] + base)
            base += int(p[key][-1])
        out[key] = np.concatenate(offs)
    return out


@pytest.mark.timeout(1500)
def test_config5_at_full_size_through_the_job_with_device_statistics(tmp_path, gpu_ctx):
    """BASELINE.json configs[4] at SURVEY's size: 3 read types x 10 000 reads of 10-50 kb, each type its own experiment
    (SAM + FASTQ) and its own model slot (hmm_0 / hmm_20 / hmm_40 = scripts/modifyHmm.py outputs), band 200, all through the
    same pipelined job; coverage / substitutions XML of each experiment from the table the job reduced on the device."""
    import multiprocessing as mp
    from nanopore_amd import ingest, job, realign as R
    from nanopore_amd.analyses.alignmentStats import MATCHES, MISMATCHES, AGAINST_N, PAIRS, SamAlignmentStats
    from nanopore_amd.analyses.coverage import LocalCoverage
    from nanopore_amd.analyses.indels import Indels
    from nanopore_amd.analyses.alignmentStats import N_INS
    from nanopore_amd.analyses.substitutions import Substitutions
    from nanopore_amd import synth
    from nanopore_amd.hmm import Hmm
    from test_gpu_stats import _count_by_hand
    n_type, parts = 10000, 10
    tmp = str(tmp_path)
    with mp.get_context("spawn").Pool(min(parts, max(2, (os.cpu_count() or 4) - 2))) as pool:
        made = pool.map(_c5_part, [(k, n_type // parts, tmp) for k in range(parts)])
    names = ("blasr_hmm_0.txt", "blasr_hmm_20.txt", "blasr_hmm_40.txt")
    types = ("2D", "template", "complement")
    ctxs = job.contexts(0, job.WORKERS)
    for c in ctxs:
        for s, nm in enumerate(names):
            c.set_hmm(Hmm.loadHmm(os.path.join(MODEL_DIR, nm)), slot=s)
    P = R.make_params(band_mode=R.BAND_FIXED, fixed_width=200)
    total_cells = 0
    for t in range(3):
        w = _concat([np.load(made[k][t]) for k in range(parts)])
        sam, fa, fq = (os.path.join(tmp, "%s.%s" % (types[t], ext)) for ext in ("sam", "fa", "fq"))
        # per-read reference slices would need 10 000 @SQ lines; the reads of one type share ONE contig made of their slices
        n = len(w["read_off"]) - 1
        w["ref_index"] = np.zeros(n, dtype=np.int32)
        w["guide_start"] = np.stack([w["ref_off"][:-1] + w["guide_start"][:, 0], np.zeros(n, dtype=np.int64)], axis=1)
        w["ref_off"] = np.array([0, len(w["ref"])], dtype=np.int64)
        synth.write_workload_files(w, sam, fa, ref_names=["contig_%s" % types[t]])
        rlen = w["read_off"][1:] - w["read_off"][:-1]
        with open(fq, "wb") as fh:                               # lengths are all the analyses take from the FASTQ
            for i in range(n):
                fh.write(b"@read_%d\n" % i + b"N" * int(rlen[i]) + b"\n+\n" + b"I" * int(rlen[i]) + b"\n")
        out = os.path.join(tmp, "%s.realigned.sam" % types[t])
        r = job.realign_sam_file(sam, out, fa, params=P, model_slot=t, want_stats=True, set_models=False)
        res, table = r["results"], r["stats"]
        assert len(res) == n_type and (res["status"] == 0).all() and rlen.max() > 45000 and rlen.min() < 12000
        assert np.allclose(res["loglik"], res["loglik_bwd"], rtol=2e-6)
        total_cells += int(res["cells"].sum())
        # the output: every record's own bytes with a new cigar that spans its window and its read
        so = ingest.SamText(out)
        fo = so.parse()
        goff, gops = so.guides(fo)
        isM, isI, isD = (gops[:, 0] == k for k in (0, 1, 2))
        csum = lambda m: np.concatenate([[0], np.cumsum(np.where(m, gops[:, 1], 0))])  # noqa: E731
        cm, ci, cd = csum(isM), csum(isI), csum(isD)
        span = lambda c: c[goff[1:]] - c[goff[:-1]]  # noqa: E731
        gw = np.concatenate([[0], np.cumsum(np.where(w["guide_ops"][:, 0] != 1, w["guide_ops"][:, 1], 0))])
        assert np.array_equal(span(cm) + span(cd), gw[w["guide_off"][1:]] - gw[w["guide_off"][:-1]]) and np.array_equal(span(cm) + span(ci), rlen)
        assert np.array_equal(fo[:, ingest.F_POS], w["guide_start"][:, 0]) and np.array_equal(r["n_ops"], goff[1:] - goff[:-1])
        # the device table: aligned pairs == the M columns of the cigars (all reads); rows == an independent per-column
        # counter on a sample (tests/test_gpu_stats.py)
        assert np.array_equal(table[:, PAIRS], span(cm)) and (table[:, 14] == 0).all()
        assert np.array_equal(table[:, MATCHES].astype(np.int64) + table[:, MISMATCHES] + table[:, AGAINST_N], table[:, PAIRS])
        contig = bytes(w["ref"]).decode()
        for i in np.argsort(rlen)[:40].tolist() + [int(np.argmax(rlen))]:
            x0 = int(w["guide_start"][i, 0])
            cig = [(int(a), int(b)) for a, b in gops[goff[i]:goff[i + 1]]]
            xs = sum(b for a, b in cig if a != 1)
            read = bytes(w["read"][w["read_off"][i]:w["read_off"][i + 1]]).decode()
            want = _count_by_hand(contig[x0:x0 + xs], read, cig, 0, 0)
            assert np.array_equal(table[i].astype(np.int64), want), (t, i)
        # the analyses' XML from that table
        stats = SamAlignmentStats.fromRealignedSam(out, fa, fq, table)
        sdir, cdir = os.path.join(tmp, "sub_%d" % t), os.path.join(tmp, "cov_%d" % t)
        os.makedirs(sdir), os.makedirs(cdir)
        sm = Substitutions(fq, types[t], fa, out, sdir).run(stats=stats)
        root = ET.parse(os.path.join(sdir, "substitutions.xml")).getroot()
        assert float(root.attrib["matches"]) == float(table[:, MATCHES].astype(np.int64).sum())
        assert float(root.attrib["mismatches"]) == float(table[:, MISMATCHES].astype(np.int64).sum())
        assert 0.7 < float(root.attrib["identity"]) < 0.99
        assert sum(sm.getCount(a, b) for a in "ACGTN" for b in "ACGTN") == float(table[:, PAIRS].astype(np.int64).sum())
        LocalCoverage(fq, types[t], fa, out, cdir).run(stats=stats)
        croot = ET.parse(os.path.join(cdir, "coverage_all.xml")).getroot()
        assert croot.attrib["numberOfReadAlignments"] == str(n_type) == croot.attrib["numberOfMappedReads"]
        ident = np.array([float(v) for v in croot.attrib["distributionidentity"].split()])
        m, x = table[:, MATCHES].astype(np.float64), table[:, MISMATCHES].astype(np.float64)
        assert np.allclose(ident, m / (m + x + table[:, 5]), rtol=1e-12)
        # indels.xml / indels.tsv (nanopore/analyses/indels.py:9-45) at this size: the gap lengths from the output's cigars (native
        # scan), their number per record cross-checked against the device table inside getAggregateIndelStats
        idir = os.path.join(tmp, "indels_%d" % t)
        os.makedirs(idir)
        Indels(fq, types[t], fa, out, idir).run(stats=stats)
        iroot = ET.parse(os.path.join(idir, "indels.xml")).getroot()
        assert iroot.attrib["numberOfReadAlignments"] == str(n_type) and len(iroot) == n_type
        nins = np.array([int(v) for v in iroot.attrib["NumberReadInsertions"].split()])
        assert nins.sum() == int(table[:, N_INS].astype(np.int64).sum()) == len(iroot.attrib["readInsertionLengths"].split())
        j = int(np.argmax(rlen))                                  # one record against its own cigar
        cig = [(int(a), int(b)) for a, b in gops[goff[j]:goff[j + 1]]]
        runs, cur = [], [0, 0]                                    # (insertion, deletion) bases between consecutive aligned blocks
        for op, ln in cig:
            if op == 0:
                runs.append(cur), None
                cur = [0, 0]
            else:
                cur[op - 1] += ln
        inner = runs[1:]                                          # (what precedes the first block is not between two pairs)
        assert iroot[j].attrib["readSeqName"] == "read_%d" % j
        assert [int(v) for v in iroot[j].attrib["readInsertionLengths"].split()] == [a for a, _ in inner if a]
        assert [int(v) for v in iroot[j].attrib["readDeletionLengths"].split()] == [b for _, b in inner if b]
        tsv = open(os.path.join(idir, "indels.tsv")).readline().split("\t")
        assert tsv[0] == "readInsertionLengths" and len(tsv) == 7
        # each type ran under ITS model: a few reads again through a plain batch with the slot's model alone
        idx = np.argsort(rlen)[:6]
        sub = synth.take_reads(w, idx)
        gpu_ctx.set_hmm(Hmm.loadHmm(os.path.join(MODEL_DIR, names[t])))
        b = gpu_ctx.stage_csr(P, sub["ref"], sub["ref_off"], sub["read"], sub["read_off"], sub["guide_ops"], sub["guide_off"],
                              ref_index=sub["ref_index"], guide_start=sub["guide_start"])
        b.run(), b.finish()
        assert np.array_equal(b.results()["loglik"], res["loglik"][idx])
        b.close()
        gpu_ctx.set_hmm(Hmm.loadHmm(HMM0))
        for path in (sam, fa, fq, out):
            os.unlink(path)
    assert total_cells > 1.5e11


def teardown_module(module):
    """The job keeps its contexts (and their device buffers) for the life of the process; the tests after this module want the memory.
    (Released, not closed: a caller's context -- the session's -- joins the pool when it is passed to realignSamFile.)"""
    from nanopore_amd import job
    for pool in job._ctx_pool.values():
        for c in pool:
            if getattr(c, "_h", None):
                c.release_scratch()
