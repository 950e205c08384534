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
