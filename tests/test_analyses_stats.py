"""Post-realign statistics: coverage / substitutions / indels analyses on a hand-built SAM
record, expected values derived by hand from the reference's definitions (coverage.py:66-85, substitutions.py:34-48,
indels.py:19-32)."""
import xml.etree.ElementTree as ET

import pytest

from nanopore_amd import bioio
from nanopore_amd.analyses.coverage import GlobalCoverage, LocalCoverage
from nanopore_amd.analyses.indels import Indels
from nanopore_amd.analyses.substitutions import Substitutions

REF = "ACGTACGTACGTACGTACGT"      # 20
#        pos 2..: 6M 2D 4M 1I 3M  against read[1:15] (1 soft-clipped base at each end)
READ = "T" + "GTACGA" + "GTAC" + "T" + "GTA" + "C"   # 16 bases: mismatch at read[6] (A vs T)


def _inputs(tmp_path, flag=0):
    fa, fq, samp = tmp_path / "ref.fa", tmp_path / "reads.fq", tmp_path / "m.sam"
    bioio.fastaWrite(str(fa), "ref1", REF)
    fq.write_text("@r1\n%s\n+\n%s\n@r2\nACGT\n+\nIIII\n" % (READ, "I" * len(READ)))
    samp.write_text("@SQ\tSN:ref1\tLN:20\n" + "\t".join(["r1", str(flag), "ref1", "3", "60", "1S6M2D4M1I3M1S", "*", "0", "0", READ, "*"]) + "\n")
    return str(fa), str(fq), str(samp)


def test_local_and_global_coverage(tmp_path):
    fa, fq, samp = _inputs(tmp_path)
    out = tmp_path / "cov"
    out.mkdir()
    LocalCoverage(fq, "2D", fa, samp, str(out)).run()
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
    GlobalCoverage(fq, "2D", fa, samp, str(out2)).run()
    rc = ET.parse(str(out2 / "coverage_all.xml")).getroot().find("readAlignmentCoverage")
    assert float(rc.attrib["readCoverage"]) == pytest.approx(13 / 16)
    assert float(rc.attrib["referenceCoverage"]) == pytest.approx(13 / 20)
    assert float(rc.attrib["insertionsPerReadBase"]) == pytest.approx(3 / 13)
    assert float(rc.attrib["deletionsPerReadBase"]) == pytest.approx(3 / 13)


def test_substitutions(tmp_path):
    fa, fq, samp = _inputs(tmp_path)
    out = tmp_path / "sub"
    out.mkdir()
    sm = Substitutions(fq, "2D", fa, samp, str(out)).run()
    root = ET.parse(str(out / "substitutions.xml")).getroot()
    assert root.attrib["matches"] == "12.0" and root.attrib["mismatches"] == "1.0"
    assert float(root.attrib["identity"]) == pytest.approx(12 / 13)
    assert sm.getCount("T", "A") == 1 and root.find("T").find("A").attrib["count"] == "1.0"
    assert [n.tag for n in root] == list("ACGTN") and [n.tag for n in root.find("A")] == list("ACGTN")
    tsv = (out / "subst.tsv").read_text().splitlines()
    assert tsv[0] == "A\tC\tG\tT" and tsv[1].split("\t")[0] == "A"
    assert [float(v) for v in tsv[4].split("\t")[1:]] == pytest.approx([0.25, 0, 0, 0.75])   # ref T: 3 matches, 1 T->A


def test_indels(tmp_path):
    fa, fq, samp = _inputs(tmp_path)
    out = tmp_path / "ind"
    out.mkdir()
    Indels(fq, "2D", fa, samp, str(out)).run()
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


def test_reverse_strand_record(tmp_path):
    """SEQ of a reverse-strand record is the reverse complement of the FASTQ read; pairs are checked base by base."""
    fa, fq, samp = tmp_path / "ref.fa", tmp_path / "reads.fq", tmp_path / "m.sam"
    bioio.fastaWrite(str(fa), "ref1", REF)
    read = bioio.reverseComplement("CGTACGTA")
    fq.write_text("@r1\n%s\n+\nIIIIIIII\n" % read)
    samp.write_text("@SQ\tSN:ref1\tLN:20\n" + "\t".join(["r1", "16", "ref1", "2", "60", "8M", "*", "0", "0", "CGTACGTA", "*"]) + "\n")
    out = tmp_path / "cov"
    out.mkdir()
    GlobalCoverage(str(fq), "2D", str(fa), str(samp), str(out)).run()
    rc = ET.parse(str(out / "coverage_all.xml")).getroot().find("readAlignmentCoverage")
    assert float(rc.attrib["identity"]) == 1.0 and float(rc.attrib["readCoverage"]) == 1.0
    assert float(rc.attrib["referenceCoverage"]) == pytest.approx(8 / 20)


def test_margin_align_base_posterior_maths():
    """calcBasePosteriorProbs / substitution matrices (marginAlignSnpCaller.py:18-35): hand-checkable cases."""
    from nanopore_amd.analyses import marginAlignSnpCaller as M
    null, flat = M.getNullSubstitutionMatrix(), M.getJukesCantorTypeSubstitutionMatrix()
    assert flat[("A", "A")] == 0.8 and flat[("A", "C")] == pytest.approx(0.2 / 3) and null[("G", "T")] == 1.0
    # all observations are 'C': posterior ~ error(missing -> C)^1, normalised
    p = M.calcBasePosteriorProbs({"A": 0.0, "C": 1.0, "G": 0.0, "T": 0.0}, "A", null, flat)
    z = 0.8 + 3 * (0.2 / 3)
    assert p["C"] == pytest.approx(0.8 / z) and p["A"] == pytest.approx((0.2 / 3) / z) and sum(p.values()) == pytest.approx(1.0)
    # an even split between A and C: symmetric posteriors for A and C
    p = M.calcBasePosteriorProbs({"A": 0.5, "C": 0.5, "G": 0.0, "T": 0.0}, "G", null, flat)
    assert p["A"] == pytest.approx(p["C"]) and p["A"] > p["G"] == pytest.approx(p["T"])
    from helpers import MODEL_DIR
    m = M.loadHmmErrorSubstitutionMatrix(MODEL_DIR + "/blasr_hmm_20.txt")
    for r in "ACGT":
        assert sum(m[(r, q)] for q in "ACGT") == pytest.approx(1.0) and m[(r, r)] > 0.5
    calls = M.SnpCalls(totalHeldOut=4)
    calls.truePositives = [(0.9, 1), (0.4, 2)]
    calls.falsePositives = [(0.95, 3), (0.1, 4)]
    assert calls.bucket([0.9, 0.4])[:41] == [2.0] * 41 and calls.bucket([0.9, 0.4])[41] == 1.0
    assert calls.getRecallByProbability()[0] == 0.5 and calls.getRecallByProbability()[95] == 0.0
    assert calls.getPrecisionByProbability()[0] == 0.5 and calls.getPrecisionByProbability()[50] == 0.5
