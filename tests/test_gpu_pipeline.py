"""End-to-end through the reference's plugin surface on the GPU (BASELINE.json configs[0], "plumbing"):
local hits -> chainSamFile -> realignSamFile (one batched C-ABI call) -> mapping.sam, then the
AlignmentUncertainty analysis, on the reference's own test data, each checked against the CPU oracle."""
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from helpers import ROOT, cigar_spans, oracle_hmm, orc
from seed_mapper import write_local_hits_sam

pytestmark = pytest.mark.gpu

C1 = os.path.join(ROOT, "tests", "golden", "c1")


def _mild_channel():
    """~94 % identity, short indels: a read the seed-and-chain step can anchor densely."""
    T = np.zeros((5, 5))
    T[0] = [0.94, 0.03, 0.03, 0.0, 0.0]
    T[1] = [0.7, 0.3, 0, 0, 0]
    T[2] = [0.7, 0, 0.3, 0, 0]
    T[3] = [1.0, 0, 0, 0, 0]
    T[4] = [1.0, 0, 0, 0, 0]
    E = np.full(80, 1.0 / 16.0)
    E[:16] = (np.full((4, 4), 0.06 / 12) + np.eye(4) * (0.235 - 0.06 / 12)).reshape(-1)
    return T.reshape(-1), E


def _slice_inputs(tmp_path, ref_span=(10000, 16000), seed=5, mild=False):
    """A 6 kb slice of the reference's test reference, and a read made from its middle 5 kb by the error
    channel of blasr_hmm_0 (the reference's own test reads are unrelated repeats: no co-linear homology)."""
    from nanopore_amd import bioio, synth
    from helpers import load_model_arrays
    rname, rseq = next(iter(bioio.fastaRead(os.path.join(C1, "reference.fa"))))
    ref = rseq[ref_span[0]:ref_span[1]].upper()
    codes = np.array(["ACGT".index(c) if c in "ACGT" else 0 for c in ref[500:5500]], dtype=np.uint8)
    T, E, _ = load_model_arrays()
    if mild:
        T, E = _mild_channel()
    rc, roff, _, _ = synth.error_channel(np.random.default_rng(seed), codes, np.array([0, len(codes)]), T, E)
    read = "".join("ACGT"[c] for c in rc)
    fq = tmp_path / "reads.fq"
    fa = tmp_path / "ref.fa"
    fq.write_text("@read_1 synthetic\n%s\n+\n%s\n" % (read, "I" * len(read)))
    bioio.fastaWrite(str(fa), rname.split()[0], ref)
    return str(fq), str(fa), {"read_1": read}, {rname.split()[0]: ref}


@pytest.mark.parametrize("mild", [True, False], ids=["dense_anchors", "sparse_anchors_wide_rectangles"])
def test_mapper_realign_matches_oracle(tmp_path, gpu_ctx, mild):
    from nanopore_amd import realign, sam as pysam
    from nanopore_amd.analyses import utils
    from nanopore_amd.analyses.alignmentUncertainty import AlignmentUncertainty
    from nanopore_amd.mappers import variants as V
    fq, fa, reads, refs = _slice_inputs(tmp_path, mild=mild)
    mapping = str(tmp_path / "mapping.sam")
    assert write_local_hits_sam(mapping, refs, reads, k=12, min_len=14) >= 10
    # keep a copy of the chained input to feed the oracle the same guide
    chained = str(tmp_path / "chained.sam")
    utils.chainSamFile(mapping, chained, fq, fa)
    mapper = V.LastParamsRealignTrainedModel(fq, "fake_readtype", fa, mapping)
    mapper.run()
    mapper.cleanup()
    out = list(pysam.Samfile(mapping, "r"))
    src = list(pysam.Samfile(chained, "r"))
    assert len(out) == len(src) == 1
    ref = next(iter(refs.values()))
    read = next(iter(reads.values()))
    for o, s in zip(out, src):
        assert (o.qname, o.pos, o.seq, o.flag) == (s.qname, 0, s.seq, s.flag)        # only the CIGAR changes (utils.py:602)
        assert cigar_spans(o.cigar) == (len(ref), len(read))
        X, Y = realign.encode(ref), realign.encode(read)
        guide = [(op, n) for op, n in s.cigar]
        P = orc.make_params(band_mode=orc.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=3000,
                            gap_gamma=0.5, match_gamma=0.0)
        m32 = orc.realign_read(oracle_hmm(), P, X, Y, guide, precision=1)
        assert m32["status"] == 0
        assert o.cigar == m32["ops"], "realigned cigar differs from the oracle (fp32 mirror)"
        m64 = orc.realign_read(oracle_hmm(), P, X, Y, guide, precision=0)
        assert o.cigar == m64["ops"], "realigned cigar differs from the fp64 oracle"
        assert o.cigar != guide                                                      # the realigner did something
    # AlignmentUncertainty on the realigned SAM (rescore mode, split 100): XML schema + oracle scores
    adir = tmp_path / "analysis_AlignmentUncertainty"
    adir.mkdir()
    an = AlignmentUncertainty(fq, "fake_readtype", fa, mapping, str(adir))
    node = an.run(ctx=gpu_ctx)
    an.cleanup()
    assert (adir / "DONE").exists()
    root = ET.parse(str(adir / "alignmentUncertainty.xml")).getroot()
    assert root.tag == "alignmentUncertainty"
    assert set(root.attrib) == {"averagePosteriorMatchProbabilityPerRead", "averagePosteriorMatchProbability",
                                "averagePosteriorMatchProbabilitesPerRead", "alignedPairsInCigar"}
    P2 = orc.make_params(band_mode=orc.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=100,
                         mode=orc.MODE_RESCORE_ORIGINAL)
    want = orc.realign_read(oracle_hmm(), P2, X, Y, [(op, n) for op, n in out[0].cigar], precision=0)
    got = float(root.attrib["averagePosteriorMatchProbabilitesPerRead"])
    assert got == pytest.approx(want["score"], abs=1e-5)
    assert int(root.attrib["alignedPairsInCigar"]) == sum(n for op, n in out[0].cigar if op == 0)
    assert float(root.attrib["averagePosteriorMatchProbability"]) == pytest.approx(got, abs=1e-12)


def test_full_size_reference_test_data_runs(tmp_path, gpu_ctx):
    """The full 32.7 kb read against the full 57.5 kb reference (wide unanchored rectangles: exercises the
    generic kernel's LDS and global rings and the split rule).  Checked through invariants only."""
    from nanopore_amd import bioio, sam as pysam
    from nanopore_amd.analyses import utils
    name, seq, _ = next(iter(bioio.fastqRead(os.path.join(C1, "reads.fq"))))
    rname, rseq = next(iter(bioio.fastaRead(os.path.join(C1, "reference.fa"))))
    fq = tmp_path / "reads.fq"
    fq.write_text("@%s\n%s\n+\n%s\n" % (name, seq, "I" * len(seq)))
    mapping = str(tmp_path / "mapping.sam")
    write_local_hits_sam(mapping, {rname.split()[0]: rseq}, {name.split()[0]: seq}, k=18, min_len=24)
    out = str(tmp_path / "realigned.sam")
    res = utils.realignSamFileTargetFn(None, mapping, out, str(fq), os.path.join(C1, "reference.fa"), 0.5, 0.0,
                                       utils.trainedModelPath("blasr_hmm_0.txt"))
    recs = list(pysam.Samfile(out, "r"))
    assert len(recs) == 1 and len(res) == 1 and res[0]["status"] == 0
    assert cigar_spans(recs[0].cigar) == (len(rseq), len(seq))
    assert res[0]["loglik"] == pytest.approx(res[0]["loglik_bwd"], rel=1e-5)
    assert res[0]["n_segments"] >= 1 and res[0]["cells"] > 1e6


def test_both_reference_test_reads_and_an_oracle_window_of_the_longer_one(tmp_path, gpu_ctx):
    """BASELINE.json configs[0] with BOTH reads of the reference's tests/readFastqFiles fixture (32 750 and 38 435 bases) against
    its 57.5 kb reference, either strand, through chainSamFile and the files -> file job.  The reads are repeats unrelated to the
    reference (no co-linear homology), so the realigner works on wide unanchored rectangles: invariants for both, and for the
    longer one a window checked against the oracle -- the matrix split (splitMatrixBiggerThanThis) makes every segment of the plan
    an independent problem, so the fp64 oracle is run on one segment the CPU finishes in seconds and its posteriors compared
    with the device's on that segment (1e-4, the tolerance north_star states)."""
    from nanopore_amd import bioio, realign as R, sam as pysam
    from nanopore_amd.analyses import utils
    from nanopore_amd.hmm import Hmm
    reads = [(n.split()[0], s) for n, s, _ in bioio.fastqRead(os.path.join(C1, "reads.fq"))]
    rname, rseq = next(iter(bioio.fastaRead(os.path.join(C1, "reference.fa"))))
    rname = rname.split()[0]
    assert [len(s) for _, s in reads] == [32750, 38435]
    fq = str(tmp_path / "reads.fq")
    with open(fq, "w") as fh:
        for n, s in reads:
            fh.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
    mapping, chained, out = (str(tmp_path / k) for k in ("mapping.sam", "chained.sam", "realigned.sam"))
    assert write_local_hits_sam(mapping, {rname: rseq}, dict(reads), k=18, min_len=24, both_strands=True) >= 2
    fa = os.path.join(C1, "reference.fa")
    utils.chainSamFile(mapping, chained, fq, fa)
    res = utils.realignSamFile(chained, out, fq, fa, utils.trainedModelPath("blasr_hmm_0.txt"), 0.5, 0.0, ctx=gpu_ctx)
    recs = list(pysam.Samfile(out, "r"))
    guides = {a.qname: a for a in pysam.Samfile(chained, "r")}
    assert sorted(a.qname for a in recs) == sorted(n for n, _ in reads) and len(res) == 2 and (res["status"] == 0).all()
    by_name = dict(reads)
    for a, r in zip(recs, res):
        assert cigar_spans(a.cigar) == (len(rseq), len(by_name[a.qname]))
        assert r["loglik"] == pytest.approx(r["loglik_bwd"], rel=1e-5) and r["cells"] > 1e6
    # the longer read: one segment of its plan against the fp64 oracle
    long_name = reads[1][0]
    g = guides[long_name]
    seq = g.seq.upper()
    X = np.array(["ACGT".index(c) if c in "ACGT" else 4 for c in rseq.upper()], dtype=np.uint8)
    Y = np.array(["ACGT".index(c) if c in "ACGT" else 4 for c in seq], dtype=np.uint8)
    guide = [(op, n) for op, n in g.cigar if op in (0, 1, 2)]
    kw = dict(band_mode=0, diagonal_expansion=10, constraint_trim=14, split_threshold=3000)
    segs = orc.plan(len(X), len(Y), guide, orc.make_params(**kw))
    pick = [s for s in segs if 2e4 <= s["cells"] <= 1.2e7]   # (a 3000 x 3000 rectangle: 9e6 cells, ~3 s of the fp64 oracle)
    assert pick, [s["cells"] for s in segs]
    seg = min(pick, key=lambda s: s["cells"])
    gpu_ctx.set_hmm(Hmm.loadHmm(utils.trainedModelPath("blasr_hmm_0.txt")))
    got = gpu_ctx.realign(R.make_params(max_pairs_per_base=24, **kw), [rseq.upper().encode()], [seq.encode()], [guide], want_pairs=True)[0]
    assert got["status"] == 0 and got["n_segments"] == len(segs)
    d = orc.fb_f64(oracle_hmm(), X[seg["xs"]:seg["xe"]], Y[seg["ys"]:seg["ye"]], seg["lo"], seg["n"], seg["ragged_start"], seg["ragged_end"], dense=False)
    want = {(int(a) + seg["xs"], int(b) + seg["ys"]): float(c) for a, b, c in zip(d["px"], d["py"], d["pp"])}
    have = {(int(a), int(b)): float(c) for a, b, c in zip(got["x"], got["y"], got["p"])
            if seg["xs"] <= a < seg["xe"] and seg["ys"] <= b < seg["ye"]}
    assert len(want) > 50
    for key in set(want) | set(have):
        assert abs(want.get(key, 0.01) - have.get(key, 0.01)) < 1e-4, key


def test_all_posteriors_mode_and_tsv(tmp_path, gpu_ctx):
    """marginAlignSnpCaller.py:136-149: --outputAllPosteriorProbs TSV = 3 numeric columns, col0 a valid
    reference index, col1 a valid index into aR.query."""
    from nanopore_amd import realign as R
    from nanopore_amd.analyses import utils
    from nanopore_amd.hmm import Hmm
    rng = np.random.default_rng(3)
    from helpers import random_pair
    X, Y, g = random_pair(rng, 500)
    gpu_ctx.set_hmm(Hmm.loadHmm(utils.trainedModelPath("blasr_hmm_20.txt")))
    P = R.make_params(band_mode=R.BAND_ANCHOR, split_threshold=100, mode=R.MODE_ALL_POSTERIORS)
    o = gpu_ctx.realign(P, [bytes(b"ACGT"[c] for c in X)], [bytes(b"ACGT"[c] for c in Y)], [g], want_pairs=True)[0]
    assert o["status"] == 0 and len(o["p"]) > 400 and cigar_spans(o["ops"]) == (len(X), len(Y))
    tsv = tmp_path / "probs.tsv"
    utils.writePosteriorProbs(str(tsv), o["x"], o["y"], o["p"])
    rows = [list(map(float, ln.split())) for ln in open(tsv)]
    assert all(len(r) == 3 and 0 <= int(r[0]) < len(X) and 0 <= int(r[1]) < len(Y) and 0.01 <= r[2] <= 1.0001 for r in rows)


def test_margin_align_snp_caller_recovers_held_out_snps(tmp_path, gpu_ctx):
    """marginAlignSnpCaller.py:40-308 in miniature: reads come from the TRUE reference, the aligner sees a MUTATED copy
    (`_Index.txt` holds the truth, :60-78); with ~40x coverage the marginalised posteriors must call most held-out SNPs."""
    from helpers import load_model_arrays
    from nanopore_amd import bioio, synth
    from nanopore_amd.analyses.marginAlignSnpCaller import MarginAlignSnpCaller
    from nanopore_amd.analyses import utils
    rng = np.random.default_rng(11)
    true = "".join("ACGT"[c] for c in rng.integers(0, 4, size=1500))
    snps = sorted(rng.choice(np.arange(50, 1450), size=12, replace=False).tolist())
    mutated = list(true)
    for i in snps:
        mutated[i] = "ACGT"[("ACGT".index(true[i]) + 1 + int(rng.integers(0, 3))) % 4]
    mutated = "".join(mutated)
    fa = tmp_path / "ref.fa"
    bioio.fastaWrite(str(fa), "chr", mutated)
    with open(str(fa) + "_Index.txt", "w") as fh:
        bioio.fastaWrite(fh, "chr", true)
        bioio.fastaWrite(fh, "chr_mutated", mutated)
    # 40 full-length reads through a mild channel; guide = true path (global, pos 0) as a chained SAM
    T, E = _mild_channel()
    codes = np.array(["ACGT".index(c) for c in true], dtype=np.uint8)
    n = 40
    off = np.arange(n + 1, dtype=np.int64) * len(codes)
    rc, roff, cols, coff = synth.error_channel(rng, np.tile(codes, n), off, T, E)
    runs, run_off = synth.columns_to_runs(cols, coff)
    fq = tmp_path / "reads.fq"
    samp = tmp_path / "mapping.sam"
    with open(fq, "w") as fqh, open(samp, "w") as sh:
        sh.write("@SQ\tSN:chr\tLN:%d\n" % len(true))
        for i in range(n):
            read = "".join("ACGT"[c] for c in rc[roff[i]:roff[i + 1]])
            fqh.write("@read%d\n%s\n+\n%s\n" % (i, read, "I" * len(read)))
            cigar = "".join("%d%s" % (l, "MID"[o]) for o, l in runs[run_off[i]:run_off[i + 1]])
            sh.write("\t".join(["read%d" % i, "0", "chr", "1", "255", cigar, "*", "0", "0", read, "*"]) + "\n")
    out = tmp_path / "analysis_MarginAlignSnpCaller"
    out.mkdir()
    an = MarginAlignSnpCaller(str(fq), "2D", str(fa), str(samp), str(out))
    an.coverages = (1000000, 10)   # keep the test short: all reads, and a 10x subsample with 3 replicates
    node = an.run(ctx=gpu_ctx, seed=5)
    an.cleanup()
    assert (out / "DONE").exists() and (out / "marginaliseConsensus.xml").exists()
    kids = list(node)
    assert len(kids) == 4 * 4 * (1 + 3)            # 4 call sets x 4 hmm types x (1 + 3 replicates)
    best = [k for k in kids if k.tag == "marginAlignMaxExpectedSnpCalls_trained_0" and k.attrib["coverage"] == "1000000"][0]
    assert best.attrib["totalHeldOut"] == "12" and best.attrib["totalSampledReads"] == "40"
    assert float(best.attrib["actualCoverage"]) > 30
    assert float(best.attrib["recall"]) >= 0.75 and float(best.attrib["precision"]) >= 0.75
    assert len(best.attrib["recallByProbability"].split()) == 101
    low = [k for k in kids if k.tag == "marginAlignMaxExpectedSnpCalls_trained_0" and k.attrib["coverage"] == "10"]
    assert len(low) == 3 and all(int(k.attrib["totalSampledReads"]) < 40 for k in low)
    gpu_ctx.set_hmm(__import__("nanopore_amd.hmm", fromlist=["Hmm"]).Hmm.loadHmm(utils.trainedModelPath("blasr_hmm_0.txt")))


def test_alignment_uncertainty_on_a_local_soft_clipped_record(tmp_path, gpu_ctx):
    """A base mapper's SAM holds LOCAL hits (pos > 0, soft clips, aend short of the reference).  The reference passes
    aR.pos / aR.aend in the exonerate cigar and cactus_realign rescoring works inside that window
    (utils.py:175-177, alignmentUncertainty.py:41); so does realignRecords: same score as the oracle on the cut-out
    window, posterior coordinates absolute."""
    from nanopore_amd import realign as R
    from nanopore_amd import sam as pysam
    from nanopore_amd.analyses.alignmentUncertainty import AlignmentUncertainty
    from nanopore_amd.analyses.utils import ANALYSIS_SPLIT_MATRIX_BIGGER_THAN, realignRecords, trainedModelPath
    fq, fa, reads, refs = _slice_inputs(tmp_path, mild=True)
    (rname, ref), read = next(iter(refs.items())), reads["read_1"]
    # the read was made from ref[500:5500]; a local hit: 7 read bases soft-clipped in front, 5 behind, gap-free guide
    n = min(len(read) - 12, 3000)
    rec = ["read_1", 0, rname, 500 + 7 + 1, 60, "7S%dM%dS" % (n, len(read) - 7 - n), "*", 0, 0, read, "*"]
    samp = tmp_path / "local.sam"
    samp.write_text("@SQ\tSN:%s\tLN:%d\n" % (rname, len(ref)) + "\t".join(str(v) for v in rec) + "\n")
    adir = tmp_path / "au"
    adir.mkdir()
    an = AlignmentUncertainty(fq, "fake_readtype", fa, str(samp), str(adir))
    an.run(ctx=gpu_ctx)
    an.cleanup()
    root = ET.parse(str(adir / "alignmentUncertainty.xml")).getroot()
    X = np.array(["ACGT".index(c) for c in ref[507:507 + n]], dtype=np.uint8)
    Y = np.array(["ACGT".index(c) for c in read[7:7 + n]], dtype=np.uint8)
    P2 = orc.make_params(band_mode=orc.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=100,
                         mode=orc.MODE_RESCORE_ORIGINAL)
    want = orc.realign_read(oracle_hmm(), P2, X, Y, [(0, n)], precision=0)
    assert float(root.attrib["averagePosteriorMatchProbabilitesPerRead"]) == pytest.approx(want["score"], abs=1e-5)
    assert int(root.attrib["alignedPairsInCigar"]) == n
    # realign mode on the same record: posteriors come back in contig / query coordinates
    sam = pysam.Samfile(str(samp), "r")
    records = list(sam)
    out = realignRecords(sam, records, refs, 0.5, 0.0, trainedModelPath("blasr_hmm_0.txt"), mode=R.MODE_ALL_POSTERIORS,
                         splitThreshold=ANALYSIS_SPLIT_MATRIX_BIGGER_THAN, ctx=gpu_ctx, want_pairs=True)
    assert out[0]["status"] == 0 and out[0]["x"].min() >= 507 and out[0]["x"].max() < 507 + n and out[0]["y"].max() < n
    assert cigar_spans(out[0]["ops"]) == (n, n)
