"""A tiny exact-match seeding 'mapper' used only by tests: stands in for last / bwa / lastz / blasr (external
binaries, absent from the reference snapshot) to produce the LOCAL hits that chainSamFile and the realigner
consume.  Writes text SAM with an @SQ header; forward strand, and with both_strands=True the reverse strand too (FLAG 16,
SEQ = the reverse complement of the read, as a mapper writes a reverse-strand hit)."""

_COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def revcomp(seq):
    return seq.translate(_COMP)[::-1]


def maximal_exact_matches(ref, read, k=16, min_len=20):
    index = {}
    for i in range(len(ref) - k + 1):
        index.setdefault(ref[i:i + k], []).append(i)
    seen = set()
    out = []
    for j in range(len(read) - k + 1):
        for i in index.get(read[j:j + k], ()):
            if (i - j, j) in seen:
                continue
            # extend left and right
            a, b = i, j
            while a > 0 and b > 0 and ref[a - 1] == read[b - 1]:
                a -= 1
                b -= 1
            e, f = i + k, j + k
            while e < len(ref) and f < len(read) and ref[e] == read[f]:
                e += 1
                f += 1
            for t in range(b, f - k + 1):
                seen.add((i - j, t))
            if e - a >= min_len:
                out.append((a, b, e - a))
    return sorted(set(out))


def write_local_hits_sam(path, refs, reads, k=16, min_len=20, both_strands=False):
    """refs, reads: dict name -> sequence.  One SAM record per maximal exact match."""
    n = 0
    with open(path, "w") as fh:
        for name, seq in refs.items():
            fh.write("@SQ\tSN:%s\tLN:%d\n" % (name, len(seq)))
        for rname, rseq in refs.items():
            R = rseq.upper()
            for qname, qseq in reads.items():
                Q = qseq.upper()
                for a, b, length in maximal_exact_matches(R, Q, k, min_len):
                    cigar = ("%dS" % b if b else "") + "%dM" % length + ("%dS" % (len(Q) - b - length) if len(Q) - b - length else "")
                    fh.write("\t".join([qname, "0", rname, str(a + 1), "255", cigar, "*", "0", "0", qseq, "*"]) + "\n")
                    n += 1
                if both_strands:
                    rc = revcomp(qseq)
                    for a, b, length in maximal_exact_matches(R, rc.upper(), k, min_len):
                        cigar = ("%dS" % b if b else "") + "%dM" % length + ("%dS" % (len(rc) - b - length) if len(rc) - b - length else "")
                        fh.write("\t".join([qname, "16", rname, str(a + 1), "255", cigar, "*", "0", "0", rc, "*"]) + "\n")
                        n += 1
    return n
