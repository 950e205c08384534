"""Post-alignment statistics on the device (include/nprealign.h: npr_align_stats,
npr_batch_align_stats): exact-integer parity with an independent test-side counter that walks the aligned pairs one by
one the way the reference's analyses do (coverage.py:36-58, substitutions.py:61-62, indels.py:21-32), and the coverage /
substitutions / indels XML / TSV schemas built from the device table, expected values derived by hand."""
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from nanopore_amd import _lib

from nanopore_amd import bioio
from nanopore_amd.analyses.coverage import GlobalCoverage, LocalCoverage
from nanopore_amd.analyses.indels import Indels
from nanopore_amd.analyses.substitutions import Substitutions

pytestmark = pytest.mark.gpu

REF = "ACGTACGTACGTACGTACGT"      # 20
#        pos 2..: 6M 2D 4M 1I 3M  against read[1:15] (1 soft-clipped base at each end)
READ = "T" + "GTACGA" + "GTAC" + "T" + "GTA" + "C"   # 16 bases: mismatch at read[6] (A vs T)


def _inputs(tmp_path, flag=0):
    fa, fq, samp = tmp_path / "ref.fa", tmp_path / "reads.fq", tmp_path / "m.sam"
    bioio.fastaWrite(str(fa), "ref1", REF)
    fq.write_text("@r1\n%s\n+\n%s\n@r2\nACGT\n+\nIIII\n" % (READ, "I" * len(READ)))
    samp.write_text("@SQ\tSN:ref1\tLN:20\n" + "\t".join(["r1", str(flag), "ref1", "3", "60", "1S6M2D4M1I3M1S", "*", "0", "0", READ, "*"]) + "\n")
    return str(fa), str(fq), str(samp)


def test_local_and_global_coverage(tmp_path, gpu_ctx):
    fa, fq, samp = _inputs(tmp_path)
    out = tmp_path / "cov"
    out.mkdir()
    LocalCoverage(fq, "2D", fa, samp, str(out)).run(ctx=gpu_ctx)
    root = ET.parse(str(out / "coverage_all.xml")).getroot()
    assert root.tag == "coverage_all" and (out / "DONE").exists() and (out / "coverage_bestPerRead.xml").exists()
    assert root.attrib["numberOfReads"] == "2" and root.attrib["numberOfMappedReads"] == "1"
    assert root.attrib["unmappedReadLengths"] == "4" and root.attrib["mappedReadLengths"] == "16"
    rc = root.find("readAlignmentCoverage")
    # 13 aligned pairs: 12 matches, 1 mismatch; one insertion of 1, one deletion of 2 (local: no end gaps)
    assert float(rc.attrib["readCoverage"]) == pytest.approx(13 / 14)
    assert float(rc.attrib["referenceCoverage"]) == pytest.approx(13 / 15)
    assert float(rc.attrib["identity"]) == pytest.approx(12 / 14)
    assert float(rc.attrib["mismatchesPerReadBase"]) == pytest.approx(1 / 13)
    assert float(rc.attrib["insertionsPerReadBase"]) == pytest.approx(1 / 13)
    assert float(rc.attrib["deletionsPerReadBase"]) == pytest.approx(1 / 13)
    assert root.attrib["avgidentity"] == rc.attrib["identity"] == root.attrib["distributionidentity"]
    lines = (out / "coverage_all.txt").read_text().splitlines()
    assert lines[0] == "MappedReadLengths 16" and lines[4].startswith("ReadIdentity 0.857")
    # global: leading/trailing unaligned read (1 + 1) and reference (2 + 3) bases count as indels
    out2 = tmp_path / "gcov"
    out2.mkdir()
    GlobalCoverage(fq, "2D", fa, samp, str(out2)).run(ctx=gpu_ctx)
    rc = ET.parse(str(out2 / "coverage_all.xml")).getroot().find("readAlignmentCoverage")
    assert float(rc.attrib["readCoverage"]) == pytest.approx(13 / 16)
    assert float(rc.attrib["referenceCoverage"]) == pytest.approx(13 / 20)
    assert float(rc.attrib["insertionsPerReadBase"]) == pytest.approx(3 / 13)
    assert float(rc.attrib["deletionsPerReadBase"]) == pytest.approx(3 / 13)


def test_substitutions(tmp_path, gpu_ctx):
    fa, fq, samp = _inputs(tmp_path)
    out = tmp_path / "sub"
    out.mkdir()
    sm = Substitutions(fq, "2D", fa, samp, str(out)).run(ctx=gpu_ctx)
    root = ET.parse(str(out / "substitutions.xml")).getroot()
    assert root.attrib["matches"] == "12.0" and root.attrib["mismatches"] == "1.0"
    assert float(root.attrib["identity"]) == pytest.approx(12 / 13)
    assert sm.getCount("T", "A") == 1 and root.find("T").find("A").attrib["count"] == "1.0"
    assert [n.tag for n in root] == list("ACGTN") and [n.tag for n in root.find("A")] == list("ACGTN")
    tsv = (out / "subst.tsv").read_text().splitlines()
    assert tsv[0] == "A\tC\tG\tT" and tsv[1].split("\t")[0] == "A"
    assert [float(v) for v in tsv[4].split("\t")[1:]] == pytest.approx([0.25, 0, 0, 0.75])   # ref T: 3 matches, 1 T->A


def test_indels(tmp_path, gpu_ctx):
    fa, fq, samp = _inputs(tmp_path)
    out = tmp_path / "ind"
    out.mkdir()
    Indels(fq, "2D", fa, samp, str(out)).run(ctx=gpu_ctx)
    root = ET.parse(str(out / "indels.xml")).getroot()
    assert root.attrib["numberOfReadAlignments"] == "1"
    assert root.attrib["readInsertionLengths"] == "1" and root.attrib["readDeletionLengths"] == "2"
    assert root.attrib["ReadSequenceLengths"] == "16" and root.attrib["NumberReadInsertions"] == "1"
    one = root.find("indels")
    assert one.attrib["numberReadDeletions"] == "1" and one.attrib["medianReadDeletionLength"] == "2.0"
    rows = [ln.split("\t") for ln in (out / "indels.tsv").read_text().splitlines()]
    assert rows[0] == ["readInsertionLengths", "readDeletionLengths", "ReadSequenceLengths", "NumberReadInsertions",
                       "NumberReadDeletions", "MedianReadInsertionLengths", "MedianReadDeletionLengths"]
    assert rows[1] == ["1", "2", "16", "1", "1", "1.0", "2.0"]


def test_reverse_strand_record(tmp_path, gpu_ctx):
    """SEQ of a reverse-strand record is the reverse complement of the FASTQ read; pairs are checked base by base."""
    fa, fq, samp = tmp_path / "ref.fa", tmp_path / "reads.fq", tmp_path / "m.sam"
    bioio.fastaWrite(str(fa), "ref1", REF)
    read = bioio.reverseComplement("CGTACGTA")
    fq.write_text("@r1\n%s\n+\nIIIIIIII\n" % read)
    samp.write_text("@SQ\tSN:ref1\tLN:20\n" + "\t".join(["r1", "16", "ref1", "2", "60", "8M", "*", "0", "0", "CGTACGTA", "*"]) + "\n")
    out = tmp_path / "cov"
    out.mkdir()
    GlobalCoverage(str(fq), "2D", str(fa), str(samp), str(out)).run(ctx=gpu_ctx)
    rc = ET.parse(str(out / "coverage_all.xml")).getroot().find("readAlignmentCoverage")
    assert float(rc.attrib["identity"]) == 1.0 and float(rc.attrib["readCoverage"]) == 1.0
    assert float(rc.attrib["referenceCoverage"]) == pytest.approx(8 / 20)




def _count_by_hand(ref, read, cigar, x0, y0):
    """Independent counter: expand the cigar into aligned pairs one by one and tally."""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    out = np.zeros(40, dtype=np.int64)
    x, y = x0, y0
    pairs, gaps = [], []           # gaps[k] = [read bases, reference bases] before aligned pair k
    pend = [0, 0]
    for op, ln in cigar:
        for _ in range(ln):
            if op == 0:
                pairs.append((x, y))
                gaps.append(pend)
                pend = [0, 0]
                x, y = x + 1, y + 1
            elif op == 1:
                pend[0] += 1
                y += 1
            else:
                pend[1] += 1
                x += 1
    for (px, py) in pairs:
        r, q = code.get(ref[px].upper(), 4), code.get(read[py].upper(), 4)
        out[15 + 5 * r + q] += 1
        if r < 4 and r == q:
            out[0] += 1
        elif r < 4 and q < 4:
            out[1] += 1
        else:
            out[2] += 1
    out[3] = len(pairs)
    for g in gaps[1:]:
        out[4] += g[0] > 0
        out[5] += g[0]
        out[6] += g[1] > 0
        out[7] += g[1]
    if pairs:
        out[8], out[9] = gaps[0]
        out[10], out[11] = pend
    else:
        out[8], out[9] = pend
    out[12], out[13] = x - x0, y - y0
    return out


def _random_alignments(rng, n):
    refs, reads, cigars, starts = [], [], [], []
    alphabet = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)
    for _ in range(n):
        kind = rng.integers(0, 5)
        cigar = []
        blocks = int(rng.integers(0, 200)) if kind else 0
        for _ in range(blocks):
            op = int(rng.choice([0, 0, 0, 1, 2]))
            ln = int(rng.integers(1, 40)) if op == 0 else int(rng.integers(0, 12))
            cigar.append((op, ln))
        if kind == 1:
            cigar = [(1, 7), (2, 3)] + cigar + [(2, 5), (1, 1)]
        if kind == 2 and cigar:
            cigar = [(0, 1)] * 70 + cigar           # more than one 64-op chunk of single-column blocks
        sx = sum(ln for op, ln in cigar if op != 1)
        sy = sum(ln for op, ln in cigar if op != 2)
        x0, y0 = int(rng.integers(0, 30)), int(rng.integers(0, 30))
        refs.append(alphabet[rng.integers(0, len(alphabet), size=x0 + sx + int(rng.integers(0, 20)))].tobytes().decode())
        reads.append(alphabet[rng.integers(0, len(alphabet), size=y0 + sy + int(rng.integers(0, 20)))].tobytes().decode())
        cigars.append(cigar)
        starts.append((x0, y0))
    return refs, reads, cigars, starts


def test_device_statistics_equal_an_independent_counter(gpu_ctx):
    rng = np.random.default_rng(2024)
    refs, reads, cigars, starts = _random_alignments(rng, 300)
    got = gpu_ctx.align_stats(refs, reads, cigars, start=starts)
    for i in range(300):
        want = _count_by_hand(refs[i], reads[i], cigars[i], *starts[i])
        assert np.array_equal(got[i].astype(np.int64), want), i
    # shared references through ref_index; a cigar that runs past its reference is flagged for that record only
    idx = [int(v) for v in rng.integers(0, 3, size=40)]
    pool = ["ACGT" * 200, "TTGCA" * 150, "N" * 700]
    cg = [[(0, int(rng.integers(1, 300))), (1, 2), (0, 100)] for _ in range(40)]
    rd = ["ACGTTGCA" * 60 for _ in range(40)]
    got = gpu_ctx.align_stats(pool, rd, cg, ref_index=idx)
    for i in range(40):
        assert np.array_equal(got[i].astype(np.int64), _count_by_hand(pool[idx[i]], rd[i], cg[i], 0, 0))
    got = gpu_ctx.align_stats(["ACGT"], ["ACGTACGT"], [[(0, 5)]])
    assert got[0, 14] == -1 and got[0, :14].sum() == 0


def test_statistics_of_a_realigned_batch_where_it_lies(gpu_ctx, monkeypatch):
    """npr_batch_align_stats: the cigars the device MEA stage just made, counted on the device without leaving it; with
    matrix splits (several segments per read); the same numbers when the cigars come from the host stage."""
    from helpers import MODEL_DIR, load_model_arrays
    from nanopore_amd import realign as R, synth
    from nanopore_amd.hmm import Hmm
    T, E, _ = load_model_arrays()
    w = synth.make_workload(77, 64, 1500, T, E, flank=0, length_sigma=0.4, len_min=200, len_max=4000)
    gpu_ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + "/blasr_hmm_0.txt"))
    for P in (R.make_params(band_mode=R.BAND_FIXED, fixed_width=100),
              R.make_params(band_mode=R.BAND_ANCHOR, constraint_trim=4, split_threshold=60, max_pairs_per_base=40)):
        tables = []
        for host_mea in (False, True):
            if host_mea:
                gpu_ctx.set_option(_lib.OPTIONS["host_mea"], 1)
            b = gpu_ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
            b.run(), b.finish()
            res, (off, ops) = b.results(), b.ops()
            tables.append(b.align_stats())
            b.close()
            gpu_ctx.set_option(_lib.OPTIONS["host_mea"], 0)
        assert (res["status"] == 0).all() and np.array_equal(tables[0], tables[1])
        if P.band_mode == R.BAND_ANCHOR:
            assert res["n_segments"].max() > 1
        for i in range(64):
            ref = bytes(w["ref"][w["ref_off"][i]:w["ref_off"][i + 1]]).decode()
            read = bytes(w["read"][w["read_off"][i]:w["read_off"][i + 1]]).decode()
            want = _count_by_hand(ref, read, [(int(a), int(c)) for a, c in ops[off[i]:off[i + 1]]], 0, 0)
            assert np.array_equal(tables[0][i].astype(np.int64), want), i


def test_posterior_scatter_add_on_the_device_equals_numpy(gpu_ctx):
    """npr_batch_base_expectations (marginAlignSnpCaller.py:150-155 on the device) against np.add.at over the same pairs,
    with a read selection, shared references and windows that start inside the reference."""
    from helpers import MODEL_DIR, load_model_arrays
    from nanopore_amd import realign as R, synth
    from nanopore_amd.hmm import Hmm
    T, E, _ = load_model_arrays()
    gpu_ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + "/blasr_hmm_0.txt"))
    w, W = synth.config_c3_shared(T, E, n_reads=96, genome_len=60000)
    reads = w["read"].copy()
    reads[::37] = ord("N")                                       # bases outside ACGT add nothing
    P = R.make_params(band_mode=R.BAND_ANCHOR, split_threshold=100, mode=R.MODE_ALL_POSTERIORS, max_pairs_per_base=48)
    b = gpu_ctx.stage_csr(P, w["ref"], w["ref_off"], reads, w["read_off"], w["guide_ops"], w["guide_off"], ref_index=w["ref_index"],
                          guide_start=w["guide_start"])
    b.run(), b.finish()
    assert (b.results()["status"] == 0).all()
    poff, px, py, pp = b.pairs()
    rng = np.random.default_rng(5)
    for use in (None, (rng.random(96) < 0.5).astype(np.uint8), np.zeros(96, dtype=np.uint8)):
        got, seen = b.base_expectations([60000], use=use)
        want = np.zeros((60000, 4))
        wseen = np.zeros(60000, dtype=bool)
        for i in range(96):
            if use is not None and not use[i]:
                continue
            x, y, p = px[poff[i]:poff[i + 1]].astype(np.int64), py[poff[i]:poff[i + 1]].astype(np.int64), pp[poff[i]:poff[i + 1]].astype(np.float64)
            code = np.array([b"ACGT".find(bytes([c])) for c in reads[w["read_off"][i] + y]])
            wseen[x] = True
            ok = code >= 0
            np.add.at(want, (x[ok], code[ok]), p[ok])
        assert np.array_equal(seen, wseen)
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, want.max()) * 50 and (got.sum() > 0) == (want.sum() > 0)
    b.close()
