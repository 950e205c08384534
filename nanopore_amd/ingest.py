"""Whole-file ingest of SAM / FASTA / FASTQ text through the native scanners of libnprealign.so (include/nprealign.h,
"bulk text ingest"; csrc/npr_io.cpp) and the splice of realigned cigars back into the records' own bytes.

The reference walks these files record by record in Python (pysam iterator + samIterator, nanopore/analyses/utils.py:287-293;
getFastaDictionary / getFastqDictionary, utils.py:233-245) and builds a Python object per record; at 50 k reads of 8 kb that
loop outlasts the DP.  Here a file is mapped once, indexed and parsed by threads, and what the batch entry points of the C
ABI need -- CSR guides, spans of the aligned part of each SEQ inside the mapped text, reference indices, window starts --
comes out as numpy arrays.  No object per record.
"""
import ctypes as C
import mmap
import os

import numpy as np

from . import _lib
from ._lib import ptr

SAM_COLS = 16
(F_QNAME_END, F_RNAME_LO, F_RNAME_HI, F_CIGAR_LO, F_CIGAR_HI, F_SEQ_LO, F_SEQ_HI, F_FLAG, F_POS, F_MAPQ, F_TID, F_QUERY_LO,
 F_QUERY_HI, F_GUIDE_OPS, F_REF_SPAN, F_STATUS) = range(SAM_COLS)
SAM_NO_REFERENCE, SAM_UNKNOWN_REFERENCE = 1, 2  # F_STATUS besides NPR_OK / NPR_ERR_INVALID (include/nprealign.h)


def _map(path):
    """The file's bytes as a read-only uint8 array (mapped; an empty file gives an empty array)."""
    if os.path.getsize(path) == 0:
        return np.zeros(0, dtype=np.uint8)
    return np.memmap(path, dtype=np.uint8, mode="r")


def _check(rc, where):
    if rc < 0:
        raise _lib.NprError(int(rc), where)
    return rc


class SamText(object):
    """A SAM file as text + line index.  `header` = the @-lines verbatim (what Samfile(..., "wh", template=sam) copies,
    utils.py:596), `references` = the @SQ names in order (sam.getrname), `span[i]` = [start, end) of alignment line i."""

    def __init__(self, path):
        L = _lib.load()
        self.path = path
        self.text = _map(path)
        n_bytes = len(self.text)
        hend = C.c_int64(0)
        # one scan when the lines are 256 bytes or more on average (reads are): the table is sized by that guess, and pages nobody
        # touches cost nothing; else count first, then fill
        guess = n_bytes // 256 + 1024
        self.span = np.zeros((guess, 2), dtype=np.int64)
        n = L.npr_sam_index(ptr(self.text), n_bytes, C.byref(hend), ptr(self.span), guess)
        if n == _lib.ERR_CAPACITY:
            n = _check(L.npr_sam_index(ptr(self.text), n_bytes, C.byref(hend), None, 0), "npr_sam_index")
            self.span = np.zeros((n, 2), dtype=np.int64)
            if n:
                _check(L.npr_sam_index(ptr(self.text), n_bytes, C.byref(hend), ptr(self.span), n), "npr_sam_index")
        else:
            self.span = self.span[:_check(n, "npr_sam_index")]
        self.header = bytes(self.text[:hend.value])
        self.references, self.lengths = [], []
        for line in self.header.decode("ascii", errors="replace").splitlines():
            if line.startswith("@SQ"):
                fields = dict(f.split(":", 1) for f in line.split("\t")[1:] if ":" in f)
                self.references.append(fields.get("SN", "*"))
                self.lengths.append(int(fields.get("LN", 0)))
        names = [r.encode() for r in self.references]
        self._rn = np.frombuffer(b"".join(names) or b"\0", dtype=np.uint8)
        self._rn_off = np.zeros(len(names) + 1, dtype=np.int64)
        if names:
            np.cumsum([len(x) for x in names], out=self._rn_off[1:])

    def __len__(self):
        return len(self.span)

    def line_lengths(self):
        return self.span[:, 1] - self.span[:, 0]

    def parse(self, lo=0, hi=None):
        """Fields of lines lo .. hi (include/nprealign.h: npr_sam_parse): int64 [hi - lo, SAM_COLS]."""
        hi = len(self.span) if hi is None else hi
        n = hi - lo
        fields = np.zeros((n, SAM_COLS), dtype=np.int64)
        if n:
            span = np.ascontiguousarray(self.span[lo:hi])
            _check(_lib.load().npr_sam_parse(ptr(self.text), ptr(span), n, ptr(self._rn), ptr(self._rn_off), len(self.references),
                                             ptr(fields)), "npr_sam_parse")
        return fields

    def records_with_a_reference(self, fields, span):
        """The mask of the lines samIterator keeps (utils.py:287-293: every record whose RNAME is not "*"), after the checks the
        reference's reader and asserts make on EVERY line: a line that does not parse, a cigar operation outside M I D S H
        (utils.py:171) or an RNAME the header does not name raises -- pysam's iterator raises on such a file; nothing
        disappears without an error."""
        status = fields[:, F_STATUS]
        bad = np.nonzero((status != 0) & (status != SAM_NO_REFERENCE))[0]
        if len(bad):
            k = int(bad[0])
            line = self.field_bytes(int(span[k, 0]), min(int(span[k, 1]), int(span[k, 0]) + 200)).decode(errors="replace")
            if status[k] == SAM_UNKNOWN_REFERENCE:
                raise KeyError("SAM record %r: RNAME %r is not among the header's @SQ lines"
                               % (line.split("\t")[0], self.field_bytes(int(fields[k, F_RNAME_LO]), int(fields[k, F_RNAME_HI])).decode(errors="replace")))
            raise AssertionError("SAM line %r: malformed, or a cigar operation outside M I D S H (utils.py:171)" % line)
        return status == 0

    def guides(self, fields, buffer=None):
        """CSR guides of parsed lines: (guide_off[n + 1], guide_ops[k, 2]) -- the M / I / D operations of each cigar, the
        operations the exonerate line carries (utils.py:173).  `buffer`: an int32 array the operations may be written into (a view
        of it is returned when it is large enough: a job's chunks then reuse one allocation of a few hundred MB)."""
        n = len(fields)
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.where(fields[:, F_STATUS] == 0, fields[:, F_GUIDE_OPS], 0), out=off[1:])
        if buffer is not None and buffer.size >= 2 * int(off[-1]):
            ops = buffer[:2 * int(off[-1])].reshape(-1, 2)
        else:
            ops = np.zeros((int(off[-1]), 2), dtype=np.int32)
        if n:
            fields = np.ascontiguousarray(fields)
            _check(_lib.load().npr_sam_guides(ptr(self.text), ptr(fields), n, ptr(off), ptr(ops)), "npr_sam_guides")
        return off, ops

    def splice(self, span, fields, word_off, n_ops, words, take=None):
        """The records with their CIGAR replaced by the packed cigars (include/nprealign.h: npr_sam_splice): uint8 array -- a view
        of the buffer `take(nbytes)` returns (a uint8 array of at least that size), when the caller has a pool of them."""
        L = _lib.load()
        n = len(fields)
        span = np.ascontiguousarray(span, dtype=np.int64)
        fields = np.ascontiguousarray(fields, dtype=np.int64)
        word_off = np.ascontiguousarray(word_off, dtype=np.int64)
        n_ops = np.ascontiguousarray(n_ops, dtype=np.int64)
        words = np.ascontiguousarray(words, dtype=np.uint32)
        rec_off = np.zeros(n + 1, dtype=np.int64)
        args = [ptr(self.text), ptr(span), ptr(fields), n, ptr(word_off), ptr(n_ops), ptr(words), ptr(rec_off)]
        if take is not None:
            # one call into a buffer sized by a bound (a record's bytes + 11 per operation: ten digits and a letter) instead of a sizing
            # call first: the sizing pass over 10^7-10^8 operations is a third of the work, and pages nobody touches cost nothing
            bound = int((span[:, 1] - span[:, 0]).sum() + 11 * n_ops.sum() + 2 * n + 1)
            out = take(bound)
            total = _check(L.npr_sam_splice(*args, ptr(out), out.nbytes), "npr_sam_splice")
            return out[:int(total)]
        total = _check(L.npr_sam_splice(*args, None, 0), "npr_sam_splice")
        out = np.empty(max(int(total), 1), dtype=np.uint8)
        _check(L.npr_sam_splice(*args, ptr(out), int(total)), "npr_sam_splice")
        return out[:int(total)]

    def field_bytes(self, lo, hi):
        return bytes(self.text[lo:hi])


class FastaTable(object):
    """getFastaDictionary (utils.py:233-238) as a table: `names` (first word of each header, unique), the sequences
    contiguous in `seq` (uint8 ASCII) with CSR offsets `off` -- the reference table npr_batch_create takes."""

    def __init__(self, path):
        L = _lib.load()
        text = _map(path)
        n = _check(L.npr_fasta_index(ptr(text), len(text), None, None, 0), "npr_fasta_index")
        rec = np.zeros((n, 4), dtype=np.int64)
        lens = np.zeros(n, dtype=np.int64)
        if n:
            _check(L.npr_fasta_index(ptr(text), len(text), ptr(rec), ptr(lens), n), "npr_fasta_index")
        self.off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=self.off[1:])
        self.seq = np.zeros(max(int(self.off[-1]), 1), dtype=np.uint8)
        if n:
            _check(L.npr_fasta_pack(ptr(text), ptr(rec), n, ptr(self.off), ptr(self.seq)), "npr_fasta_pack")
        self.names = [bytes(text[a:b]).decode("ascii", errors="replace") for a, b in rec[:, :2]]
        self.index = {}
        for k, name in enumerate(self.names):
            assert name not in self.index, "Duplicate fasta sequence name %s" % name  # utils.py:236
            self.index[name] = k

    def __len__(self):
        return len(self.names)

    def sequence(self, name):
        k = self.index[name]
        return bytes(self.seq[self.off[k]:self.off[k + 1]]).decode("ascii")


def fastq_table(path):
    """getFastqDictionary (utils.py:240-245) without a Python loop over the lines: (names, text, seq spans [n, 2])."""
    L = _lib.load()
    text = _map(path)
    n = _check(L.npr_fastq_index(ptr(text), len(text), None, 0), "npr_fastq_index")
    rec = np.zeros((n, 4), dtype=np.int64)
    if n:
        _check(L.npr_fastq_index(ptr(text), len(text), ptr(rec), n), "npr_fastq_index")
    names = [bytes(text[a:b]).decode("ascii", errors="replace") for a, b in rec[:, :2]]
    return names, text, rec[:, 2:4].copy()
