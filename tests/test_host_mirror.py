"""Host-side mirror of the reference's interface on the path (no GPU): SAM / FASTA / FASTQ / exonerate-cigar
plumbing, model selection truth table, chaining, mapper class surface.  Expected values are hand-derived
from the reference's format strings and asserts (SURVEY.md 8c "Python callers / harness")."""
import os

import pytest

from nanopore_amd import bioio, sam as pysam
from nanopore_amd.analyses import utils
from nanopore_amd.mappers import variants as V
from nanopore_amd.mappers.abstractMapper import AbstractMapper

REF = "ACGTACGTTAGGCTAGCTAGGATCGATCGTAGCTAGCTAGGCTAGCTAACG"           # 51 bases
READ = "ACGTACGTTAGGCTAGCAGGATCGATCGTTTAGCTAGCTAGGCTAGCTAACG"         # 1 deletion, 2 inserted T


def _write_inputs(tmp_path, records, refs=None):
    fa = tmp_path / "ref.fa"
    fq = tmp_path / "reads.fq"
    refs = refs or {"ref1": REF}
    with open(fa, "w") as fh:
        for n, s in refs.items():
            bioio.fastaWrite(fh, n + " some description", s)
    with open(fq, "w") as fh:
        fh.write("@read1 extra\n%s\n+\n%s\n" % (READ, "I" * len(READ)))
    samp = tmp_path / "in.sam"
    with open(samp, "w") as fh:
        fh.write("@HD\tVN:1.0\n")
        for n, s in refs.items():
            fh.write("@SQ\tSN:%s\tLN:%d\n" % (n, len(s)))
        for r in records:
            fh.write("\t".join(str(v) for v in r) + "\n")
    return str(fa), str(fq), str(samp)


def test_fasta_fastq_round_trip(tmp_path):
    fa, fq, _ = _write_inputs(tmp_path, [])
    assert utils.getFastaDictionary(fa) == {"ref1": REF}
    assert utils.getFastqDictionary(fq) == {"read1": READ}
    assert list(bioio.fastqRead(fq)) == [("read1 extra", READ, "I" * len(READ))]
    assert bioio.reverseComplement("ACGTN") == "NACGT"
    assert bioio.nameValue("loadHmm", None) == "" and bioio.nameValue("loadHmm", "f.txt") == "--loadHmm=f.txt"


def test_sam_record_attributes():
    a = pysam.AlignedRead()
    a.cigar = pysam.parseCigar("2H3S10M2I5M3D4M1S")
    a.seq = "N" * (3 + 10 + 2 + 5 + 4 + 1)
    a.pos = 7
    assert (a.qstart, a.qend, len(a.query)) == (3, 24, 21)
    assert a.aend == 7 + 10 + 5 + 3 + 4 and a.alen == 22
    pairs = a.aligned_pairs
    assert pairs[0] == (0, 7) and (10, None) in pairs and (None, 22) in pairs   # positions index `query`
    assert sum(1 for q, r in pairs if q is not None and r is not None) == 19
    assert pysam.formatCigar(a.cigar) == "2H3S10M2I5M3D4M1S"
    with pytest.raises(RuntimeError):
        pysam.parseCigar("10M5")


def test_exonerate_cigar_string(tmp_path):
    """format utils.py:175-177: query first (0 .. qend-qstart), target second (pos .. aend), score literal 1;
    soft and hard clips dropped."""
    rec = ["read1", 0, "ref1", 3, 60, "2S10M1D5M2I8M1S", "*", 0, 0, "A" * 28, "*"]
    fa, fq, samp = _write_inputs(tmp_path, [rec])
    sam = pysam.Samfile(samp, "r")
    aR = next(iter(sam))
    line = utils.getExonerateCigarFormatString(aR, sam)
    assert line == "cigar: read1 0 25 + ref1 2 26 + 1 M 10 D 1 M 5 I 2 M 8"
    pA = bioio.cigarReadFromString(line)
    assert (pA.contig2, pA.start2, pA.end2, pA.contig1, pA.start1, pA.end1, pA.score) == ("read1", 0, 25, "ref1", 2, 26, 1.0)
    assert [(o.type, o.length) for o in pA.operationList] == [(0, 10), (2, 1), (0, 5), (1, 2), (0, 8)]
    assert bioio.cigarToString(pA) == line
    with pytest.raises(RuntimeError):
        bioio.cigarReadFromString("cigar: read1 0 25 + ref1 2 26 + 1 M 10")   # ops do not span the coordinates
    # reverse strand record: still '+' in the line (SEQ is already reverse-complemented, utils.py:327-331)
    rec2 = ["read1", 16, "ref1", 1, 60, "5M", "*", 0, 0, "ACGTA", "*"]
    _, _, samp2 = _write_inputs(tmp_path, [rec2])
    sam2 = pysam.Samfile(samp2, "r")
    a2 = next(iter(sam2))
    assert a2.is_reverse and utils.getExonerateCigarFormatString(a2, sam2) == "cigar: read1 0 5 + ref1 0 5 + 1 M 5"


def test_sam_iterator_drops_unmapped_and_writer_copies_header(tmp_path):
    recs = [["read1", 0, "ref1", 1, 60, "5M", "*", 0, 0, "ACGTA", "*", "NM:i:0"],
            ["read2", 4, "*", 0, 0, "*", "*", 0, 0, "ACGTA", "*"]]
    fa, fq, samp = _write_inputs(tmp_path, recs)
    sam = pysam.Samfile(samp, "r")
    got = list(utils.samIterator(sam))
    assert [a.qname for a in got] == ["read1"] and got[0].tags == ["NM:i:0"]
    out = pysam.Samfile(str(tmp_path / "out.sam"), "wh", template=sam)
    got[0].cigar = [(0, 2), (1, 1), (0, 2)]
    out.write(got[0])
    out.close()
    lines = open(tmp_path / "out.sam").read().splitlines()
    assert lines[:2] == ["@HD\tVN:1.0", "@SQ\tSN:ref1\tLN:%d" % len(REF)]
    assert lines[2].split("\t")[:6] == ["read1", "0", "ref1", "1", "60", "2M1I2M"]


def test_chain_sam_file_produces_global_records(tmp_path):
    # two co-linear local hits of read1 on ref1 and a third, far-away, lower-scoring one
    recs = [["read1", 0, "ref1", 1, 60, "17M35S", "*", 0, 0, READ, "*"],
            ["read1", 0, "ref1", 19, 60, "17S11M2I22M", "*", 0, 0, READ, "*"],
            ["read1", 0, "ref1", 40, 60, "48S4M", "*", 0, 0, READ, "*"]]
    fa, fq, samp = _write_inputs(tmp_path, recs)
    out = str(tmp_path / "chained.sam")
    utils.chainSamFile(samp, out, fq, fa)
    sam = pysam.Samfile(out, "r")
    got = list(sam)
    assert len(got) == 1
    c = got[0]
    assert c.pos == 0 and c.seq == READ and not c.is_reverse                    # utils.py:313-331
    assert c.aend == len(REF) and c.qstart == 0 and c.qend == len(READ)         # utils.py:492-496
    assert sum(n for op, n in c.cigar if op in (0, 2)) == len(REF)              # utils.py:381
    assert sum(n for op, n in c.cigar if op in (0, 1)) == len(READ)             # utils.py:382
    assert c.cigar == [(0, 17), (2, 1), (0, 11), (1, 2), (0, 22)]


def test_model_selection_truth_table(tmp_path):
    m = AbstractMapper("reads.fq", "2D", "ref.fa", "out.sam", emptyHmmFile="/x/hmm.txt")
    assert m.selectHmmFile() is None                                           # stock model, abstractMapper.py:36-37
    assert m.selectHmmFile(doEm=True) == "/x/hmm.txt"                          # :32-33
    p = m.selectHmmFile(useTrainedModel=True)
    assert p.endswith(os.path.join("mappers", "blasr_hmm_0.txt")) and os.path.exists(p)   # :34-35
    assert m.selectHmmFile(useTrainedModel=True, trainedModelFile="blasr_hmm_40.txt").endswith("blasr_hmm_40.txt")
    with pytest.raises(RuntimeError, match="Attempting to train stock model"):
        m.selectHmmFile(doEm=True, useTrainedModel=True)                        # :29-30
    m.cleanup()


def test_mapper_class_surface():
    # names are part of the output directory layout (pipeline.py:107-108)
    for name in ("LastParamsRealign", "LastParamsRealignEm", "LastParamsRealignTrainedModel",
                 "LastParamsRealignTrainedModel20", "LastParamsRealignTrainedModel40", "BwaParamsRealign",
                 "BlasrParamsRealignTrainedModel40", "LastzParamsRealignEm", "CombinedMapperRealign", "BwaChain"):
        cls = getattr(V, name)
        assert issubclass(cls, AbstractMapper) and cls.__name__ == name
    m = V.LastParamsRealign("r.fq", "2D", "ref.fa", "/nonexistent/mapping.sam")
    with pytest.raises(RuntimeError, match="run the external mapper first"):
        m.run()
    m.cleanup()


def test_em_m_step_and_xml(tmp_path):
    """M-step = renormalised expected counts; the XML has the schema analyses/hmm.py reads."""
    import xml.etree.ElementTree as ET
    import numpy as np
    from nanopore_amd import em
    from nanopore_amd.hmm import Hmm
    rng = np.random.default_rng(1)
    h = em.randomise(Hmm(), rng)
    T = np.array(h.transitions).reshape(5, 5)
    assert np.allclose(T.sum(axis=1), 1.0) and int((T > 0).sum()) == 15
    assert np.allclose(np.array(h.emissions).reshape(5, 16).sum(axis=1), 1.0)
    Texp = np.zeros(25)
    Texp[[0, 1, 2]] = [6.0, 3.0, 1.0]          # only the match row observed
    Eexp = np.zeros(80)
    Eexp[:16] = np.arange(16)
    before = list(h.transitions)
    em.normalise(h, Texp, Eexp, trainEmissions=True)
    assert h.transitions[:5] == [0.6, 0.3, 0.1, 0.0, 0.0]
    assert h.transitions[5:] == before[5:]      # rows without counts keep their values
    assert np.allclose(h.emissions[:16], np.arange(16) / 120.0)
    h.likelihood = -12.5
    xml = tmp_path / "hmm.txt.xml"
    em.writeXML(str(xml), [h, h.copy()], [[-20.0, -13.0, -12.5], [-19.0, -12.5]])
    root = ET.parse(str(xml)).getroot()
    assert len(root.findall("transition")) == 25 and len(root.findall("emission")) == 80 and len(root.findall("hmm")) == 2
    t = root.findall("transition")[1]
    assert (t.attrib["from"], t.attrib["to"]) == ("0", "1") and float(t.attrib["avg"]) == pytest.approx(0.3) and float(t.attrib["std"]) == 0.0
    e = [x for x in root.findall("emission") if x.attrib["state"] == "0"][5]
    assert (e.attrib["x"], e.attrib["y"]) == ("C", "C")
    assert [float(v) for v in root.findall("hmm")[0].attrib["runningLikelihoods"].split()] == [-20.0, -13.0, -12.5]
    o = em.Options()                              # option names of utils.py:509-523
    assert (o.modelType, o.trials, o.iterations, o.randomStart, o.trainEmissions) == ("fiveStateAsymmetric", 3, 100, True, True)
    assert "--splitMatrixBiggerThanThis=300" in o.optionsToRealign
