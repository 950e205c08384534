"""The `cactus_realign` executable (nanopore_amd/csrc/cactus_realign_main.cpp: a host main() over the C ABI) run with the
reference's literal call strings -- nanopore/analyses/utils.py:586-587, alignmentUncertainty.py:41,
marginAlignSnpCaller.py:136-146 -- and compared with the results of the C ABI itself."""
import os
import subprocess

import numpy as np
import pytest

from helpers import MODEL_DIR, ROOT, random_pair

pytestmark = pytest.mark.gpu

EXE = os.path.join(ROOT, "nanopore_amd", "cactus_realign")


def _inputs(tmp_path, n=5):
    from nanopore_amd import bioio
    rng = np.random.default_rng(4242)
    cases = [random_pair(rng, int(rng.integers(200, 900)), indel=0.15, max_indel=6) for _ in range(n)]
    refs = {"ref%d" % i: "".join("ACGT"[c] for c in X) for i, (X, _, _) in enumerate(cases)}
    reads = {"read%d" % i: "".join("ACGT"[c] for c in Y) for i, (_, Y, _) in enumerate(cases)}
    fa, rd = str(tmp_path / "ref.fa"), str(tmp_path / "read.fa")
    with open(fa, "w") as fh:
        for k, v in refs.items():
            bioio.fastaWrite(fh, k + " some description", v)
    with open(rd, "w") as fh:
        for k, v in reads.items():
            bioio.fastaWrite(fh, k, v)
    lines = []
    for i, (X, Y, g) in enumerate(cases):
        lines.append("cigar: read%d 0 %d + ref%d 0 %d + 1 %s" % (i, len(Y), i, len(X), " ".join("%s %d" % ("MID"[op], ln) for op, ln in g)))
    return fa, rd, refs, reads, cases, lines


def _run(cmd, stdin):
    p = subprocess.run(cmd, shell=True, input=stdin.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout.decode().splitlines()


def test_the_reference_call_strings(tmp_path, gpu_ctx):
    from nanopore_amd import bioio, realign as R
    from nanopore_amd.hmm import Hmm
    assert os.path.exists(EXE), "build it: make -C nanopore_amd/csrc"
    fa, rd, refs, reads, cases, lines = _inputs(tmp_path)
    hmm = os.path.join(MODEL_DIR, "blasr_hmm_0.txt")
    gpu_ctx.set_hmm(Hmm.loadHmm(hmm))
    rl = [refs["ref%d" % i] for i in range(len(cases))]
    ql = [reads["read%d" % i] for i in range(len(cases))]
    gl = [g for _, _, g in cases]

    # utils.py:586-587 -- one cigar per process in the reference, any number here
    cmd = "%s %s %s --diagonalExpansion=10 --splitMatrixBiggerThanThis=3000 %s --gapGamma=%s --matchGamma=%s" % (
        EXE, fa, rd, bioio.nameValue("loadHmm", hmm), 0.5, 0.0)
    out = _run(cmd, "\n".join(lines) + "\n")
    want = gpu_ctx.realign(R.make_params(band_mode=R.BAND_ANCHOR, diagonal_expansion=10, split_threshold=3000, gap_gamma=0.5, match_gamma=0.0),
                           rl, ql, gl)
    assert len(out) == len(lines)                                              # exactly one cigar per cigar (utils.py:588-589)
    for i, line in enumerate(out):
        pA = bioio.cigarReadFromString(line)
        assert (pA.contig2, pA.start2, pA.end2, pA.contig1, pA.start1, pA.end1) == ("read%d" % i, 0, len(ql[i]), "ref%d" % i, 0, len(rl[i]))
        assert [(o.type, o.length) for o in pA.operationList] == want[i]["ops"]
        assert pA.score == pytest.approx(want[i]["score"], abs=1e-6)
    # one cigar at a time, as the reference pipes them, and no --loadHmm (stock model)
    one = _run("echo %s | %s %s %s --diagonalExpansion=10 --splitMatrixBiggerThanThis=3000 %s --gapGamma=0.5 --matchGamma=0.0" % (
        "'" + lines[2] + "'", EXE, fa, rd, bioio.nameValue("loadHmm", None)), "")
    gpu_ctx.set_hmm(None)
    w2 = gpu_ctx.realign(R.make_params(band_mode=R.BAND_ANCHOR), rl[2:3], ql[2:3], gl[2:3])[0]
    assert [(o.type, o.length) for o in bioio.cigarReadFromString(one[0]).operationList] == w2["ops"]
    gpu_ctx.set_hmm(Hmm.loadHmm(hmm))

    # alignmentUncertainty.py:41
    post = str(tmp_path / "post.tsv")
    cmd = "cat %s | %s %s %s --rescoreByPosteriorProbIgnoringGaps --rescoreOriginalAlignment --diagonalExpansion=10 " \
          "--splitMatrixBiggerThanThis=100 --outputPosteriorProbs=%s --loadHmm=%s" % (str(tmp_path / "in.cig"), EXE, fa, rd, post, hmm)
    (tmp_path / "in.cig").write_text("\n".join(lines) + "\n")
    out = _run(cmd, "")
    want = gpu_ctx.realign(R.make_params(band_mode=R.BAND_ANCHOR, diagonal_expansion=10, split_threshold=100, mode=R.MODE_RESCORE_ORIGINAL),
                           rl, ql, gl, want_pairs=True)
    n_rows = 0
    for i, line in enumerate(out):
        pA = bioio.cigarReadFromString(line)
        assert [(o.type, o.length) for o in pA.operationList] == [(op, ln) for op, ln in gl[i] if ln > 0]   # ops verbatim (:51-52)
        assert pA.score == pytest.approx(want[i]["score"], abs=1e-6)
        have = {(int(a), int(b)) for a, b in zip(want[i]["x"], want[i]["y"])}
        x = y = 0
        for op, ln in gl[i]:
            if op == 0:
                n_rows += sum(1 for t in range(ln) if (x + t, y + t) in have)
                x, y = x + ln, y + ln
            elif op == 1:
                y += ln
            else:
                x += ln
    rows = [ln.split() for ln in open(post).read().splitlines()]
    assert len(rows) == n_rows and all(len(r) == 3 and 0.01 <= float(r[2]) <= 1.0001 for r in rows)

    # marginAlignSnpCaller.py:136-146
    allp = str(tmp_path / "all.tsv")
    cmd = "echo '%s' | %s %s %s --diagonalExpansion=10 --splitMatrixBiggerThanThis=100 --outputAllPosteriorProbs=%s --loadHmm=%s" % (
        lines[1], EXE, fa, rd, allp, hmm)
    out = _run(cmd, "")
    w = gpu_ctx.realign(R.make_params(band_mode=R.BAND_ANCHOR, diagonal_expansion=10, split_threshold=100, mode=R.MODE_ALL_POSTERIORS),
                        rl[1:2], ql[1:2], gl[1:2], want_pairs=True)[0]
    assert [(o.type, o.length) for o in bioio.cigarReadFromString(out[0]).operationList] == w["ops"]
    rows = np.array([[float(v) for v in ln.split()] for ln in open(allp).read().splitlines()])   # parsed as :149 parses them
    assert np.array_equal(rows[:, 0].astype(np.int64), w["x"]) and np.array_equal(rows[:, 1].astype(np.int64), w["y"])
    assert np.array_equal(rows[:, 2].astype(np.float32), w["p"])


def test_failures_exit_non_zero(tmp_path):
    fa, rd, refs, reads, cases, lines = _inputs(tmp_path, n=2)
    for stdin in ("cigar: nobody 0 5 + ref0 0 5 + 1 M 5\n", "cigar: read0 0 5 + ref0 0 7 + 1 M 5\n", "not a cigar\n"):
        p = subprocess.run([EXE, fa, rd, "--diagonalExpansion=10"], input=stdin.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert p.returncode != 0 and p.stderr


def test_exonerate_lines_of_chained_records_through_the_executable(tmp_path, gpu_ctx):
    """SURVEY.md 8a row a3 on the GPU path: real chained records -- a forward read inside its reference (leading / trailing
    D), a reverse-strand read, a read with unaligned bases at both ends (leading / trailing I) -- go through
    getExonerateCigarFormatString (utils.py:168-180), the line is piped into the cactus_realign executable exactly as
    realignCigarTargetFn does (utils.py:576-589: reference and aR.query as FASTA files, the cigar on stdin), and the cigar
    that comes back is the one realignRecords gives for the same records."""
    from nanopore_amd import bioio, sam as pysam
    from nanopore_amd.analyses import utils
    from seed_mapper import revcomp, write_local_hits_sam
    rng = np.random.default_rng(99)
    dna = lambda n: "".join("ACGT"[c] for c in rng.integers(0, 4, size=n))  # noqa: E731

    def noisy(seq):
        X = np.array(["ACGT".index(c) for c in seq], dtype=np.uint8)
        out, x = [], 0
        while x < len(X):
            r = rng.random()
            if r < 0.02:
                x += int(rng.integers(1, 3))
            elif r < 0.04:
                out.extend(rng.integers(0, 4, size=int(rng.integers(1, 3))).tolist())
            else:
                out.append(int(X[x]) if rng.random() > 0.05 else int(rng.integers(0, 4)))
                x += 1
        return "".join("ACGT"[c] for c in out)

    refs = {"refA": dna(2400), "refB": dna(1800)}
    reads = {"fwd_inside": noisy(refs["refA"][400:1900]),                       # leading and trailing D
             "rev_strand": revcomp(noisy(refs["refB"][200:1500])),             # FLAG 16
             "ragged_ends": dna(60) + noisy(refs["refA"][0:1200]) + dna(45)}   # leading and trailing I (the read overhangs)
    fa, fq = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fq")
    with open(fa, "w") as fh:
        for k, v in refs.items():
            bioio.fastaWrite(fh, k, v)
    with open(fq, "w") as fh:
        for k, v in reads.items():
            fh.write("@%s\n%s\n+\n%s\n" % (k, v, "I" * len(v)))
    hits, chained = str(tmp_path / "hits.sam"), str(tmp_path / "chained.sam")
    assert write_local_hits_sam(hits, refs, reads, k=12, min_len=16, both_strands=True) >= 9
    utils.chainSamFile(hits, chained, fq, fa)
    sam = pysam.Samfile(chained, "r")
    records = list(utils.samIterator(sam))
    assert sorted(aR.qname for aR in records) == sorted(reads)
    by_name = {aR.qname: aR for aR in records}
    assert by_name["rev_strand"].is_reverse and not by_name["fwd_inside"].is_reverse
    assert by_name["fwd_inside"].cigar[0][0] == 2 and by_name["fwd_inside"].cigar[0][1] >= 380    # the reference before the read: a leading D
    assert {op for op, _ in by_name["fwd_inside"].cigar[-2:]} <= {1, 2}                            # ... and what the chain left unaligned behind it
    assert 1 in {op for op, _ in by_name["ragged_ends"].cigar[:2]} and by_name["ragged_ends"].cigar[-1][0] == 1  # overhanging read: I at both ends
    hmm = os.path.join(MODEL_DIR, "blasr_hmm_0.txt")
    want = utils.realignRecords(sam, records, utils.getFastaDictionary(fa), 0.5, 0.0, hmm, ctx=gpu_ctx)
    for aR, w in zip(records, want):
        line = utils.getExonerateCigarFormatString(aR, sam)
        assert line.startswith("cigar: %s 0 %d + %s 0 %d + 1 " % (aR.qname, len(aR.query), sam.getrname(aR.rname), len(refs[sam.getrname(aR.rname)])))
        # realignCigarTargetFn's temp files: the whole reference sequence, and aR.query under the read's name
        tref, tread = str(tmp_path / "ref_one.fa"), str(tmp_path / "read_one.fa")
        bioio.fastaWrite(tref, sam.getrname(aR.rname), refs[sam.getrname(aR.rname)])
        bioio.fastaWrite(tread, aR.qname, aR.query)
        out = _run("echo %s | %s %s %s --diagonalExpansion=10 --splitMatrixBiggerThanThis=3000 %s --gapGamma=%s --matchGamma=%s" % (
            line, EXE, tref, tread, bioio.nameValue("loadHmm", hmm), 0.5, 0.0), "")
        assert len(out) == 1                                                    # utils.py:588-589
        pA = bioio.cigarReadFromString(out[0])
        assert w["status"] == 0 and [(o.type, o.length) for o in pA.operationList] == w["ops"]
        assert pA.score == pytest.approx(w["score"], abs=1e-6)
        assert w["ops"] != [(op, n) for op, n in aR.cigar]                     # the realigner moved something
