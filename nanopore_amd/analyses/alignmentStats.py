"""Per-record alignment statistics of a SAM file, reduced on the GPU.

The reference's coverage / substitutions / indels analyses (nanopore/analyses/coverage.py:10-95,
substitutions.py:9-56, indels.py:9-45) each walk every aligned pair of every record in Python.  Here the records go to
the device once (`npr_align_stats`, include/nprealign.h: one wavefront per record reduces its cigar and bases to a row of
integers: matches, mismatches, pairs against N, gaps between aligned pairs, the 5 x 5 substitution counts) and the three
analyses format their XML / TSV schemas from that table.  Only what the integer table cannot hold -- the individual gap
lengths indels.xml lists -- is taken from the cigars on the host, vectorised.  No CPU fallback: the device does the counting.
"""
import numpy as np

from .. import sam as pysam
from .utils import clipLengths, getFastaDictionary, getFastqDictionary, samIterator

# columns of the device table (include/nprealign.h: NPR_STATS_WORDS)
MATCHES, MISMATCHES, AGAINST_N, PAIRS, N_INS, INS_LEN, N_DEL, DEL_LEN, LEAD_READ, LEAD_REF, TRAIL_READ, TRAIL_REF, REF_SPAN, \
    READ_SPAN, STATUS = range(15)
SUBST = 15  # 25 counts, reference base major, A C G T N


class _Lengths(object):
    """Names -> sequence lengths with the dict surface the analyses use on refSequences / readSequences (len(), items()
    yielding something with a len()): what a job-sized read set needs instead of 30 k Python strings of 30 kb."""

    class _Sized(object):
        __slots__ = ("n",)

        def __init__(self, n):
            self.n = int(n)

        def __len__(self):
            return self.n

    def __init__(self, names, lengths):
        self._len = {}
        for name, n in zip(names, lengths):
            assert name not in self._len, "Duplicate sequence name %s" % name
            self._len[name] = int(n)

    def __len__(self):
        return len(self._len)

    def __contains__(self, name):
        return name in self._len

    def length(self, name):
        return self._len[name]

    def items(self):
        return ((k, _Lengths._Sized(v)) for k, v in self._len.items())


class SamAlignmentStats(object):
    """The records of a SAM file (those with a reference, utils.samIterator) and their device-reduced statistics."""

    def __init__(self, samFile, referenceFastaFile, readFastqFile, ctx=None):
        from .utils import _context
        self.refSequences = getFastaDictionary(referenceFastaFile)
        self.readSequences = getFastqDictionary(readFastqFile)
        sam = pysam.Samfile(samFile, "r")
        records = list(samIterator(sam))
        self.readNames = [aR.qname for aR in records]
        self.refNames = [sam.getrname(aR.rname) for aR in records]
        self.isReverse = np.array([aR.is_reverse for aR in records], dtype=bool)
        self.pos = np.array([aR.pos for aR in records], dtype=np.int64)
        self.aend = np.array([aR.aend for aR in records], dtype=np.int64)
        self.refLength = np.array([len(self.refSequences[r]) for r in self.refNames], dtype=np.int64)
        self.readLength = np.array([len(self.readSequences[q]) for q in self.readNames], dtype=np.int64)
        clips = [clipLengths(aR) for aR in records]
        self.clipBefore = np.array([c[0] for c in clips], dtype=np.int64)
        self.clipAfter = np.array([c[1] for c in clips], dtype=np.int64)
        self.cigars = [[(op, ln) for op, ln in aR.cigar if op in (0, 1, 2)] for aR in records]
        for aR in records:
            assert all(op in (0, 1, 2, 4, 5) for op, _ in aR.cigar), "unsupported cigar operation in %s" % aR.qname
        names = sorted(self.refSequences)
        index = {n: i for i, n in enumerate(names)}
        n = len(records)
        if n:
            ctx = ctx or _context()
            self.table = ctx.align_stats([self.refSequences[k] for k in names], [aR.query for aR in records], self.cigars,
                                         ref_index=[index[r] for r in self.refNames], start=[(int(p), 0) for p in self.pos])
            bad = np.flatnonzero(self.table[:, STATUS] != 0)
            if len(bad):
                raise RuntimeError("The cigar of record %s runs past its sequences" % self.readNames[int(bad[0])])
        else:
            self.table = np.zeros((0, 40), dtype=np.int32)
        sam.close()
        self._gaps = None

    @classmethod
    def fromRealignedSam(cls, samFile, referenceFastaFile, readFastqFile, table):
        """The same object for the output of a realignment job, from the table the job already reduced on the device where
        the alignments lay (job.realign_sam_file(..., want_stats=True): npr_batch_align_stats per chunk) -- no second pass over
        the aligned pairs, and no Python object per record: names, positions and lengths come from the native scanners
        (nanopore_amd/ingest.py).  Row i of `table` belongs to record i of `samFile` (the records with a reference, in order).
        The individual gap lengths (indels.xml) come from the cigars of `samFile`, scanned natively."""
        from .. import ingest
        self = cls.__new__(cls)
        sam = ingest.SamText(samFile)
        f = sam.parse()
        kept = np.nonzero(sam.records_with_a_reference(f, sam.span))[0]
        f = f[kept]
        if len(f) != len(table):
            raise ValueError("%d records with a reference in %s, %d rows of statistics" % (len(f), samFile, len(table)))
        fasta = ingest.FastaTable(referenceFastaFile)
        qnames, qtext, qspan = ingest.fastq_table(readFastqFile)
        self.refSequences = _Lengths(fasta.names, fasta.off[1:] - fasta.off[:-1])
        self.readSequences = _Lengths(qnames, qspan[:, 1] - qspan[:, 0])
        starts = sam.span[kept, 0]
        self.readNames = [sam.field_bytes(int(a), int(b)).decode() for a, b in zip(starts, f[:, ingest.F_QNAME_END])]
        self.refNames = [sam.references[int(t)] for t in f[:, ingest.F_TID]]
        self.isReverse = (f[:, ingest.F_FLAG] & 0x10) != 0
        self.pos = f[:, ingest.F_POS].copy()
        self.aend = self.pos + f[:, ingest.F_REF_SPAN]
        self.refLength = np.array([self.refSequences.length(r) for r in self.refNames], dtype=np.int64)
        self.readLength = np.array([self.readSequences.length(q) for q in self.readNames], dtype=np.int64)
        # clips of the record as it stands in the file: SEQ outside the aligned part
        self.clipBefore = f[:, ingest.F_QUERY_LO] - f[:, ingest.F_SEQ_LO]
        self.clipAfter = f[:, ingest.F_SEQ_HI] - f[:, ingest.F_QUERY_HI]
        self.cigars = None
        self._cigar_csr = sam.guides(f)  # (offsets, (op, length) of the M / I / D operations): what the indel lengths are read from
        self.table = np.ascontiguousarray(table, dtype=np.int32)
        bad = np.flatnonzero(self.table[:, STATUS] != 0)
        if len(bad):
            raise RuntimeError("The cigar of record %s runs past its sequences" % self.readNames[int(bad[0])])
        self._gaps = None
        return self

    def __len__(self):
        return len(self.readNames)

    def gapLengths(self):
        """(insertion lengths, deletion lengths) per record, in alignment order: read / reference bases between two
        consecutive aligned pairs (indels.py:22-25).  Vectorised over all cigar operations of the file."""
        if self._gaps is None:
            n = len(self)
            ins, dels = [[] for _ in range(n)], [[] for _ in range(n)]
            if self.cigars is not None:
                counts = np.array([len(c) for c in self.cigars], dtype=np.int64)
                ops = np.array([o for c in self.cigars for o in c], dtype=np.int64).reshape(-1, 2) if counts.sum() else np.zeros((0, 2), dtype=np.int64)
            else:  # fromRealignedSam: the cigars' M / I / D operations as the native scanner left them (CSR)
                off, csr = self._cigar_csr
                counts, ops = (off[1:] - off[:-1]).astype(np.int64), csr.astype(np.int64)
            if counts.sum():
                rec = np.repeat(np.arange(n), counts)
                first = np.concatenate([[0], np.cumsum(counts)[:-1]])
                is_m = (ops[:, 0] == 0) & (ops[:, 1] > 0)
                before = np.cumsum(is_m) - is_m                      # aligned blocks before this op, file-wide
                gap = before - np.repeat(before[np.minimum(first, len(before) - 1)], counts)  # ... within its record: 0 = before the first block
                blocks = np.bincount(rec, weights=is_m, minlength=n).astype(np.int64)
                inner = (~is_m) & (gap > 0) & (gap < blocks[rec])
                span = int(gap.max()) + 1
                for code, out in ((1, ins), (2, dels)):
                    sel = inner & (ops[:, 0] == code) & (ops[:, 1] > 0)
                    key = rec[sel] * span + gap[sel]
                    uniq, inv = np.unique(key, return_inverse=True)
                    total = np.bincount(inv, weights=ops[sel, 1]).astype(np.int64)
                    owner = uniq // span                                # (sorted: a record's gaps are contiguous, in alignment order)
                    cuts = np.searchsorted(owner, np.arange(n + 1))
                    for k in np.flatnonzero(cuts[1:] > cuts[:-1]):
                        out[int(k)] = total[cuts[k]:cuts[k + 1]].tolist()
            self._gaps = (ins, dels)
        return self._gaps

    def indelTotals(self, globalAlignment):
        """(number of read insertions, their total length, number of read deletions, their total length) per record.
        A global alignment also counts what lies before the first and after the last aligned pair: clipped and inserted
        read bases, and the reference from its first base to its last (coverage.py:42-58)."""
        t = self.table.astype(np.int64)
        n_ins, ins_len, n_del, del_len = t[:, N_INS].copy(), t[:, INS_LEN].copy(), t[:, N_DEL].copy(), t[:, DEL_LEN].copy()
        if globalAlignment:
            has_pairs = t[:, PAIRS] > 0
            for extra_read, extra_ref in ((self.clipBefore + t[:, LEAD_READ], self.pos + t[:, LEAD_REF]),
                                          (self.clipAfter + t[:, TRAIL_READ], self.refLength - self.aend + t[:, TRAIL_REF])):
                n_ins += (extra_read > 0) & has_pairs
                ins_len += np.where(has_pairs, extra_read, 0)
                n_del += (extra_ref > 0) & has_pairs
                del_len += np.where(has_pairs, extra_ref, 0)
        return n_ins, ins_len, n_del, del_len

    def substitutionCounts(self):
        """5 x 5 counts of aligned (reference base, read base), A C G T N, summed over the file."""
        return self.table[:, SUBST:SUBST + 25].astype(np.int64).sum(axis=0).reshape(5, 5)
