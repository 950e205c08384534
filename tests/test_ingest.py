"""Bulk text ingest / splice (include/nprealign.h "bulk text ingest", csrc/npr_io.cpp; host code, no GPU): the native
scanners against the record-at-a-time host mirror (nanopore_amd/sam.py, bioio.py) on hand-made and random files -- the
fields realignSamFile2TargetFn reads per record (nanopore/analyses/utils.py:563-570) and the output realignSamFile3TargetFn
writes (utils.py:591-609)."""
import os

import numpy as np
import pytest

from nanopore_amd import bioio, ingest, sam as pysam

EDGE = [
    "@HD\tVN:1.0",
    "@SQ\tSN:chr1\tLN:1000",
    "@SQ\tSN:chr2\tLN:500",
    "@PG\tID:x",
    "r1\t0\tchr1\t11\t60\t2S5M1I3M2D4M3S\t*\t0\t0\tACGTACGTACGTACGTAC\tIIIIIIIIIIIIIIIIII\tNM:i:3\tXX:Z:foo",
    "r2\t16\tchr2\t1\t255\t3H4M2H\t*\t0\t0\tACGT\t*",
    "r3\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\t*",                    # unmapped: samIterator drops it
    "r4\t0\tchrUn\t5\t1\t4M\t*\t0\t0\tACGT\t*",               # RNAME not in the header: tid -1, flagged (SAM_UNKNOWN_REFERENCE)
    "",
    "r5\t0\tchr1\t100\t3\t10M\t=\t7\t-3\tACGTACGTAC\tJJJJJJJJJJ",
    "r6\t0\tchr1\t1\t3\t1H2S3M1D2I1M1S4H\t*\t0\t0\tACGTACGTA\t*",
]


def _write(tmp_path, lines, name="a.sam", eol="\n", final=True):
    p = str(tmp_path / name)
    with open(p, "w", newline="") as fh:
        fh.write(eol.join(lines) + (eol if final else ""))
    return p


def _check_against_mirror(path):
    st = ingest.SamText(path)
    f = st.parse()
    recs = list(pysam.Samfile(path, "r"))
    assert len(recs) == len(st) == len(f)
    assert st.references == pysam.Samfile(path, "r").references
    for i, a in enumerate(recs):
        assert a.rname == f[i, ingest.F_TID]
        assert st.field_bytes(int(st.span[i, 0]), int(f[i, ingest.F_QNAME_END])).decode() == a.qname
        if a.rname < 0:   # "*": the record samIterator drops; any other name: an error of the file, never dropped silently
            assert f[i, ingest.F_STATUS] == (ingest.SAM_NO_REFERENCE if a._rname == "*" else ingest.SAM_UNKNOWN_REFERENCE)
        elif a.cigar:
            assert f[i, ingest.F_STATUS] == 0
            assert st.field_bytes(int(f[i, ingest.F_QUERY_LO]), int(f[i, ingest.F_QUERY_HI])).decode() == (a.query or "")
            assert (f[i, ingest.F_POS], f[i, ingest.F_FLAG], f[i, ingest.F_MAPQ]) == (a.pos, a.flag, a.mapq)
            assert f[i, ingest.F_REF_SPAN] == a.aend - a.pos
            assert f[i, ingest.F_GUIDE_OPS] == sum(1 for op, _ in a.cigar if op in (0, 1, 2))
    keep = f[:, ingest.F_TID] >= 0
    ff, sp = f[keep], st.span[keep]
    off, ops = st.guides(ff)
    kept = [a for a in recs if a.rname != -1]
    for i, a in enumerate(kept):
        assert [tuple(int(v) for v in r) for r in ops[off[i]:off[i + 1]]] == [(op, n) for op, n in a.cigar if op in (0, 1, 2)]
    return st, ff, sp, kept


@pytest.mark.parametrize("eol,final", [("\n", True), ("\n", False), ("\r\n", True)])
def test_parse_matches_the_record_reader_on_edge_cases(tmp_path, eol, final):
    _check_against_mirror(_write(tmp_path, EDGE, eol=eol, final=final))


def test_splice_equals_the_record_writer(tmp_path):
    path = _write(tmp_path, EDGE)
    st, ff, sp, kept = _check_against_mirror(path)
    rng = np.random.default_rng(3)
    new = [[(int(rng.integers(0, 3)), int(rng.integers(1, 2000))) for _ in range(int(rng.integers(0, 6)))] for _ in kept]
    words = np.array([(n << 2) | op for c in new for op, n in c], dtype=np.uint32)
    nops = np.array([len(c) for c in new], dtype=np.int64)
    woff = np.concatenate([[0], np.cumsum(nops)[:-1]]).astype(np.int64)
    got = st.header + bytes(st.splice(sp, ff, woff, nops, words))
    # ... and the same bytes in one call into a buffer sized by a bound (what the job's pooled buffers get); long operations too
    pooled = st.splice(sp, ff, woff, nops, words, take=lambda nbytes: np.full(nbytes + 5, 0x55, dtype=np.uint8))
    assert st.header + bytes(pooled) == got
    big = words.copy()
    big[::3] = (np.array([9, 10, 99, 100, 12345, (1 << 29) - 1], dtype=np.uint32)[np.arange(len(big[::3])) % 6] << 2) | (big[::3] & 3)
    assert bytes(st.splice(sp, ff, woff, nops, big, take=lambda nbytes: np.empty(nbytes, dtype=np.uint8))) == bytes(st.splice(sp, ff, woff, nops, big))
    text = bytes(st.splice(sp, ff, woff, nops, big)).decode()
    assert all(("%d%s" % (int(w) >> 2, "MID"[int(w) & 3])) in text for w in big[:12])
    out = str(tmp_path / "b.sam")
    src = pysam.Samfile(path, "r")
    dst = pysam.Samfile(out, "wh", template=src)
    for a, c in zip((a for a in src if a.rname != -1), new):
        a.cigar = c                                             # realignSamFile3TargetFn: only the cigar changes (utils.py:602)
        dst.write(a)
    dst.close()
    assert got == open(out, "rb").read()


def test_malformed_records_are_flagged_not_skipped(tmp_path):
    lines = EDGE[:3] + ["bad1\t0\tchr1\t5\t1\t4M2N3M\t*\t0\t0\tACGTACG\t*",     # N: outside M I D S H (utils.py:171)
                        "bad2\t0\tchr1\t5\t1\t4M\t*\t0",                        # too few columns
                        "bad3\tx\tchr1\t5\t1\t4M\t*\t0\t0\tACGT\t*",            # FLAG not a number
                        "bad4\t0\tchr1\t5\t1\t4\t*\t0\t0\tACGT\t*",             # cigar without an operation
                        "bad5\t0\tchr1\t5\t1\t*\t*\t0\t0\tACGT\t*",             # mapped, no cigar
                        "bad6\t0\tchr1\t5\t1\t99999999999M\t*\t0\t0\tACGT\t*",   # eleven digits
                        "bad7\t0\tchr1\t5\t1\t4294967296M\t*\t0\t0\tACGT\t*",    # ten digits, beyond 2^29
                        "bad8\t0\tchr1\t5\t1\t" + "9" * 40 + "M\t*\t0\t0\tACGT\t*",  # a number that wraps 64 bits
                        "bad9\t0\tchr1\t5\t1\tM4\t*\t0\t0\tACGT\t*",             # a letter without a number
                        "bad10\t0\tchr1\t5\t1\t2M2m\t*\t0\t0\tACGT\t*",          # a letter that is no operation
                        "ok\t0\tchr1\t5\t1\t4M\t*\t0\t0\tACGT\t*",
                        "ok2\t0\tchr1\t5\t1\t2S536870911M1I2D3H\t*\t0\t0\tACGT\t*"]   # the largest length there is
    st = ingest.SamText(_write(tmp_path, lines))
    f = st.parse()
    assert list(f[:, ingest.F_STATUS]) == [-1] * 10 + [0, 0]
    assert f[11, ingest.F_GUIDE_OPS] == 3 and f[11, ingest.F_REF_SPAN] == 536870911 + 2
    off, ops = st.guides(f)
    assert ops.tolist() == [[0, 4], [0, 536870911], [1, 1], [2, 2]] and off.tolist() == [0] * 11 + [1, 4]


def test_nothing_is_dropped_without_an_error(tmp_path):
    """Only RNAME "*" is dropped (samIterator, utils.py:287-293).  A line that does not parse, or an RNAME the header does not
    name (a SAM without its @SQ lines, a truncated file), raises before any record is filtered -- pysam's iterator raises on
    such files; a successful job with fewer records would be silent data loss."""
    head, recs = EDGE[:4], [EDGE[4], EDGE[5], EDGE[6], EDGE[9], EDGE[10]]
    st = ingest.SamText(_write(tmp_path, head + recs))
    f = st.parse()
    keep = st.records_with_a_reference(f, st.span)
    assert list(keep) == [True, True, False, True, True] and list(f[:, ingest.F_STATUS]) == [0, 0, ingest.SAM_NO_REFERENCE, 0, 0]
    for bad, exc in (("r4\t0\tchrUn\t5\t1\t4M\t*\t0\t0\tACGT\t*", KeyError),         # RNAME missing from the header
                     ("no tabs at all", AssertionError),
                     ("r7\tzz\tchr1\t5\t1\t4M\t*\t0\t0\tACGT\t*", AssertionError),      # FLAG not a number
                     ("r8\t0\tchr1\t5\t1\t4M\t*\t0\t0\tAC", AssertionError)):            # truncated in the middle of a record
        st = ingest.SamText(_write(tmp_path, head + recs + [bad], name="bad.sam"))
        f = st.parse()
        with pytest.raises(exc):
            st.records_with_a_reference(f, st.span)
    # a file without its @SQ lines: every mapped record names an unknown reference
    st = ingest.SamText(_write(tmp_path, ["@HD\tVN:1.0"] + recs, name="nosq.sam"))
    with pytest.raises(KeyError):
        st.records_with_a_reference(st.parse(), st.span)


def test_random_sam_files(tmp_path):
    rng = np.random.default_rng(11)
    for case in range(5):
        refs = ["ctg%d" % k for k in range(int(rng.integers(1, 5)))]
        lines = ["@HD\tVN:1.0\tSO:unsorted"] + ["@SQ\tSN:%s\tLN:%d" % (r, 10000) for r in refs]
        for i in range(int(rng.integers(1, 400))):
            ops = []
            if rng.random() < 0.3:
                ops.append((5, int(rng.integers(1, 30))))
            if rng.random() < 0.4:
                ops.append((4, int(rng.integers(1, 30))))
            for _ in range(int(rng.integers(1, 12))):
                ops.append((int(rng.integers(0, 3)), int(rng.integers(1, 40))))
            if rng.random() < 0.4:
                ops.append((4, int(rng.integers(1, 30))))
            if rng.random() < 0.3:
                ops.append((5, int(rng.integers(1, 30))))
            qlen = sum(n for op, n in ops if op in (0, 1, 4))
            seq = "".join("ACGTN"[c] for c in rng.integers(0, 5, size=qlen)) or "*"
            rname = refs[int(rng.integers(0, len(refs)))] if rng.random() < 0.9 else "*"
            tags = ["NM:i:%d" % i] if rng.random() < 0.5 else []
            lines.append("\t".join(["q%d" % i, str(int(rng.choice([0, 16, 256, 272]))), rname, str(int(rng.integers(1, 9000))),
                                    str(int(rng.integers(0, 256))), pysam.formatCigar(ops), "*", "0", "0", seq,
                                    "*" if rng.random() < 0.5 or seq == "*" else "I" * len(seq)] + tags))
        _check_against_mirror(_write(tmp_path, lines, name="r%d.sam" % case))


def test_empty_and_header_only_files(tmp_path):
    for name, lines in (("e.sam", []), ("h.sam", EDGE[:4])):
        p = str(tmp_path / name)
        open(p, "w").write("\n".join(lines) + ("\n" if lines else ""))
        st = ingest.SamText(p)
        assert len(st) == 0 and st.parse().shape == (0, ingest.SAM_COLS)
        assert st.references == [l.split("\t")[1][3:] for l in lines if l.startswith("@SQ")]


def test_fasta_and_fastq_tables(tmp_path):
    from helpers import ROOT
    c1 = os.path.join(ROOT, "tests", "golden", "c1")
    fa = ingest.FastaTable(os.path.join(c1, "reference.fa"))
    want = [(n.split()[0], s) for n, s in bioio.fastaRead(os.path.join(c1, "reference.fa"))]
    assert fa.names == [n for n, _ in want] and [fa.sequence(n) for n in fa.names] == [s for _, s in want]
    names, text, spans = ingest.fastq_table(os.path.join(c1, "reads.fq"))
    wantq = [(n.split()[0], s) for n, s, _ in bioio.fastqRead(os.path.join(c1, "reads.fq"))]
    assert names == [n for n, _ in wantq]
    assert [bytes(text[a:b]).decode() for a, b in spans] == [s for _, s in wantq]
    # ragged FASTA: blank lines, trailing blanks, CRLF, a record without sequence, no newline at the end
    p = str(tmp_path / "x.fa")
    open(p, "w", newline="").write(">a desc here\r\nACGT  \r\n\r\nAC\r\n>b\n>c\tz\nNNNN\nac")
    t = ingest.FastaTable(p)
    want = [(n.split()[0], s) for n, s in bioio.fastaRead(p)]
    assert [(n, t.sequence(n)) for n in t.names] == want == [("a", "ACGTAC"), ("b", ""), ("c", "NNNNac")]
    with pytest.raises(AssertionError):
        open(p, "w").write(">a\nAC\n>a\nGT\n")
        ingest.FastaTable(p)                                    # duplicate names (utils.py:236)


@pytest.mark.parametrize("windowed", [True, False])
def test_a_workload_written_as_files_reads_back_as_the_same_arrays(tmp_path, windowed):
    """synth.write_workload_files (the bench's and the GPU tests' input files) -> SamText / FastaTable -> the arrays the job
    stages: reads where they lie in the text, CSR guides, window starts, reference table."""
    from helpers import load_model_arrays
    from nanopore_amd import job, synth
    T, E, _ = load_model_arrays()
    if windowed:
        w, _ = synth.config_c3_shared(T, E, n_reads=40, genome_len=50000)
    else:
        w = synth.make_workload(3, 25, 600, T, E, flank=50)
    sam, fa, fq = (str(tmp_path / k) for k in ("a.sam", "a.fa", "a.fq"))
    synth.write_workload_files(w, sam, fa, fastq_path=fq)
    st, table = ingest.SamText(sam), ingest.FastaTable(fa)
    f = st.parse()
    src = job.SamSource(st, table, st.span, f)
    n = len(w["read_off"]) - 1
    assert src.n == n and (f[:, ingest.F_STATUS] == 0).all()
    assert np.array_equal(src.guide_off, w["guide_off"]) and np.array_equal(src.guide_ops, w["guide_ops"])
    assert np.array_equal(src.ref, np.asarray(w["ref"])) and np.array_equal(src.ref_off, w["ref_off"])
    for i in range(n):
        assert bytes(src.text[src.read_begin[i]:src.read_end[i]]) == bytes(w["read"][w["read_off"][i]:w["read_off"][i + 1]])
    if windowed:
        assert np.array_equal(src.guide_start, w["guide_start"]) and (src.ref_index == 0).all()
    else:
        assert (src.guide_start == 0).all() and np.array_equal(src.ref_index, np.arange(n))
    # chunking: contiguous, covering, balanced by bases
    chunks = job.chunk_bounds(src.lengths(), 0, n, chunk_bases=int(src.lengths().sum() // 4), workers=2)
    assert chunks[0][0] == 0 and chunks[-1][1] == n and all(a[1] == b[0] for a, b in zip(chunks, chunks[1:])) and 3 <= len(chunks) <= 5
    # the record-at-a-time mirror sees the same records
    recs = list(pysam.Samfile(sam, "r"))
    assert [a.qname for a in recs] == ["read_%d" % i for i in range(n)]
    assert [a.query.encode() for a in recs] == [bytes(w["read"][w["read_off"][i]:w["read_off"][i + 1]]) for i in range(n)]
    assert [s for _, s, _ in bioio.fastqRead(fq)] == [a.seq for a in recs]


def test_many_short_lines_take_the_counting_path(tmp_path):
    """SamText sizes its line table by a guess (lines of 256 bytes or more: one scan) and counts first when the guess is too small:
    4000 records of 40 bytes, against the record reader."""
    lines = EDGE[:4] + ["s%d\t0\tchr1\t%d\t60\t4M\t*\t0\t0\tACGT\tIIII" % (i, 1 + i % 900) for i in range(4000)]
    path = _write(tmp_path, lines, name="short.sam")
    assert os.path.getsize(path) // 256 + 1024 < 4000
    st, ff, sp, kept = _check_against_mirror(path)
    assert len(st) == 4000 and (st.line_lengths() >= 30).all()
