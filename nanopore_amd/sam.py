"""Minimal text-SAM reader / writer exposing the pysam-0.7 attribute surface the realign path relies on
(pysam is not installable here; SURVEY.md Appendix B lists each attribute and where the reference uses it).

Only what nanopore/analyses/utils.py:168-180, :540-609, alignmentUncertainty.py and marginAlignSnpCaller.py touch
is implemented: header pass-through, record iteration in file order, `cigar` get/set, soft/hard-clip aware
`query/qstart/qend`, `pos/aend`, `aligned_pairs`, `is_reverse/is_unmapped`, `getrname`.
"""

_OPS = "MIDNSHP=X"
_CODE = {c: i for i, c in enumerate(_OPS)}


def parseCigar(text):
    if text == "*" or not text:
        return None
    out, num = [], 0
    seen = False
    for ch in text:
        if ch.isdigit():
            num = num * 10 + ord(ch) - 48
            seen = True
        else:
            if ch not in _CODE or not seen:
                raise RuntimeError("Malformed CIGAR %r" % text)
            out.append((_CODE[ch], num))
            num, seen = 0, False
    if seen:
        raise RuntimeError("Malformed CIGAR %r" % text)
    return out


def formatCigar(cigar):
    if not cigar:
        return "*"
    return "".join("%i%s" % (length, _OPS[op]) for op, length in cigar)


class AlignedRead(object):
    __slots__ = ("qname", "flag", "_rname", "rname", "pos", "mapq", "cigar", "rnext", "pnext", "tlen", "seq",
                 "qual", "tags")

    def __init__(self):
        self.qname = "*"
        self.flag = 0
        self._rname = "*"
        self.rname = -1   # tid; -1 = no reference (samIterator drops these, utils.py:287-293)
        self.pos = -1     # 0-based
        self.mapq = 255
        self.cigar = None
        self.rnext = "*"
        self.pnext = 0
        self.tlen = 0
        self.seq = None
        self.qual = None
        self.tags = []

    @property
    def is_reverse(self):
        return bool(self.flag & 0x10)

    @property
    def is_unmapped(self):
        return bool(self.flag & 0x4)

    @property
    def cigarstring(self):
        return formatCigar(self.cigar)

    @property
    def qstart(self):
        """Offset in SEQ of the first base that is not soft-clipped."""
        n = 0
        for op, length in self.cigar or []:
            if op == 4:
                n += length
            elif op == 5:
                continue
            else:
                break
        return n

    @property
    def qend(self):
        if self.seq is None:
            return 0
        n = len(self.seq)
        for op, length in reversed(self.cigar or []):
            if op == 4:
                n -= length
            elif op == 5:
                continue
            else:
                break
        return n

    @property
    def query(self):
        """SEQ without the soft-clipped ends: what goes to the realigner (utils.py:570)."""
        return None if self.seq is None else self.seq[self.qstart:self.qend]

    @property
    def alen(self):
        return sum(length for op, length in self.cigar or [] if op in (0, 2, 3, 7, 8))

    @property
    def aend(self):
        """One past the last aligned reference position."""
        return self.pos + self.alen if self.cigar else None

    @property
    def aligned_pairs(self):
        """[(read position or None, reference position or None)].  As in pysam 0.7, the API the reference was
        written against, read positions count from the first NON-clipped base, i.e. they index `query`
        (the reference indexes alignedRead.query[readPos] at utils.py:150 and adds qstart itself in
        getAbsoluteReadOffset, utils.py:157-166)."""
        out = []
        q, r = 0, self.pos
        for op, length in self.cigar or []:
            if op in (0, 7, 8):
                out.extend((q + i, r + i) for i in range(length))
                q += length
                r += length
            elif op == 1:
                out.extend((q + i, None) for i in range(length))
                q += length
            elif op in (2, 3):
                out.extend((None, r + i) for i in range(length))
                r += length
        return out


class Samfile(object):
    """Samfile(path, "r") to read; Samfile(path, "wh", template=other) to write text SAM with the header
    copied from `template` (utils.py:456, :596)."""

    def __init__(self, path, mode="r", template=None):
        self.path = path
        self.mode = mode
        self.header_lines = []
        self.references = []
        self.lengths = []
        if mode == "r":
            self._fh = open(path)
            self._pending = None
            while True:
                pos = self._fh.tell()
                line = self._fh.readline()
                if not line:
                    break
                if line.startswith("@"):
                    self.header_lines.append(line.rstrip("\r\n"))
                    if line.startswith("@SQ"):
                        fields = dict(f.split(":", 1) for f in line.rstrip("\r\n").split("\t")[1:] if ":" in f)
                        self.references.append(fields.get("SN", "*"))
                        self.lengths.append(int(fields.get("LN", 0)))
                else:
                    self._fh.seek(pos)
                    break
            self._tid = {n: i for i, n in enumerate(self.references)}
        elif mode in ("w", "wh"):
            self._fh = open(path, "w")
            if template is not None:
                self.header_lines = list(template.header_lines)
                self.references = list(template.references)
                self.lengths = list(template.lengths)
            self._tid = {n: i for i, n in enumerate(self.references)}
            if mode == "wh":
                for h in self.header_lines:
                    self._fh.write(h + "\n")
        else:
            raise ValueError("Unsupported mode %r" % mode)

    def getrname(self, tid):
        return self.references[tid]

    def gettid(self, name):
        return self._tid.get(name, -1)

    def __iter__(self):
        return self

    def __next__(self):
        while True:
            line = self._fh.readline()
            if not line:
                raise StopIteration
            line = line.rstrip("\r\n")
            if not line or line.startswith("@"):
                continue
            f = line.split("\t")
            if len(f) < 11:
                raise RuntimeError("Malformed SAM record: %r" % line)
            a = AlignedRead()
            a.qname = f[0]
            a.flag = int(f[1])
            a._rname = f[2]
            a.rname = self._tid.get(f[2], -1)
            a.pos = int(f[3]) - 1
            a.mapq = int(f[4])
            a.cigar = parseCigar(f[5])
            a.rnext = f[6]
            a.pnext = int(f[7])
            a.tlen = int(f[8])
            a.seq = None if f[9] == "*" else f[9]
            a.qual = None if f[10] == "*" else f[10]
            a.tags = f[11:]
            return a

    next = __next__

    def write(self, a):
        rname = self.references[a.rname] if 0 <= a.rname < len(self.references) else "*"
        self._fh.write("\t".join([a.qname, str(a.flag), rname, str(a.pos + 1), str(a.mapq), formatCigar(a.cigar),
                                  a.rnext, str(a.pnext), str(a.tlen), a.seq if a.seq is not None else "*",
                                  a.qual if a.qual is not None else "*"] + list(a.tags)) + "\n")

    def close(self):
        if self._fh:
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
