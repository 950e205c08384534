"""Small stand-ins for the sonLib.bioio helpers the realign path uses (sonLib is an empty submodule in the
reference snapshot): FASTA/FASTQ readers and writer, the exonerate-cigar `PairwiseAlignment` type with its
reader/writer, `nameValue` and a temp-dir `Target` shim.  SURVEY.md Appendix A/B pin the behaviour.

Cigar text grammar (nanopore/analyses/utils.py:173-177):
    cigar: <query> <qstart> <qend> <+|-> <target> <tstart> <tend> <+|-> <score> (M|I|D <len>)*
Operation types carry the SAM numbering (M 0, I 1, D 2): realignSamFile3TargetFn copies `op.type` straight
into `aR.cigar` (utils.py:602).
"""
import os
import shutil
import tempfile


def fastaRead(fileHandleOrFile):
    """Yields (name, sequence); the name is the full header line without '>'."""
    fh = open(fileHandleOrFile) if isinstance(fileHandleOrFile, str) else fileHandleOrFile
    try:
        name, chunks = None, []
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    yield name, "".join(chunks)
                name, chunks = line[1:], []
            elif line and name is not None:
                chunks.append(line.strip())
        if name is not None:
            yield name, "".join(chunks)
    finally:
        if isinstance(fileHandleOrFile, str):
            fh.close()


def fastaWrite(fileHandleOrFile, name, seq, mode="w"):
    fh = open(fileHandleOrFile, mode) if isinstance(fileHandleOrFile, str) else fileHandleOrFile
    try:
        fh.write(">%s\n" % name)
        for i in range(0, len(seq), 100):
            fh.write(seq[i:i + 100] + "\n")
    finally:
        if isinstance(fileHandleOrFile, str):
            fh.close()


def fastqRead(fileHandleOrFile):
    """Yields (name, sequence, quality string or None); four-line records."""
    fh = open(fileHandleOrFile) if isinstance(fileHandleOrFile, str) else fileHandleOrFile
    try:
        while True:
            head = fh.readline()
            if not head:
                break
            head = head.rstrip("\r\n")
            if not head:
                continue
            if not head.startswith("@"):
                raise RuntimeError("Malformed FASTQ record header: %r" % head)
            seq = fh.readline().rstrip("\r\n")
            plus = fh.readline().rstrip("\r\n")
            qual = fh.readline().rstrip("\r\n")
            if not plus.startswith("+"):
                raise RuntimeError("Malformed FASTQ record for %s" % head)
            yield head[1:], seq, (qual if qual else None)
    finally:
        if isinstance(fileHandleOrFile, str):
            fh.close()


def reverseComplement(seq):
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "a": "t", "c": "g", "g": "c", "t": "a", "N": "N", "n": "n"}
    return "".join(comp.get(c, c) for c in reversed(seq))


def nameValue(name, value, valueType=str):
    """'--name=value', or '' when value is None (sonLib semantics; utils.py:586)."""
    if value is None:
        return ""
    if valueType is bool:
        return "--%s" % name if value else ""
    return "--%s=%s" % (name, str(value))


class AlignmentOperation(object):
    def __init__(self, opType, length, score=0.0):
        self.type = opType
        self.length = length
        self.score = score

    def __eq__(self, other):
        return self.type == other.type and self.length == other.length


class PairwiseAlignment(object):
    PAIRWISE_MATCH = 0     # 'M'
    PAIRWISE_INDEL_Y = 1   # 'I': read (query) only
    PAIRWISE_INDEL_X = 2   # 'D': reference (target) only
    PAIRWISE_PLUS = "+"
    PAIRWISE_MINUS = "-"
    _LETTER = {0: "M", 1: "I", 2: "D"}
    _CODE = {"M": 0, "I": 1, "D": 2}

    def __init__(self, contig1, start1, end1, strand1, contig2, start2, end2, strand2, score, operationList):
        # contig1 = target (reference), contig2 = query (read): SURVEY.md Appendix A naming
        self.contig1, self.start1, self.end1, self.strand1 = contig1, start1, end1, strand1
        self.contig2, self.start2, self.end2, self.strand2 = contig2, start2, end2, strand2
        self.score = score
        self.operationList = operationList


def cigarReadFromString(line):
    tok = line.split()
    if len(tok) < 10 or tok[0] != "cigar:":
        raise RuntimeError("Malformed cigar line: %r" % line)
    ops = []
    rest = tok[10:]
    if len(rest) % 2:
        raise RuntimeError("Malformed cigar operations: %r" % line)
    for letter, length in zip(rest[::2], rest[1::2]):
        if letter not in PairwiseAlignment._CODE:
            raise RuntimeError("Unknown cigar operation %r" % letter)
        ops.append(AlignmentOperation(PairwiseAlignment._CODE[letter], int(length)))
    pA = PairwiseAlignment(tok[5], int(tok[6]), int(tok[7]), tok[8] == "+", tok[1], int(tok[2]), int(tok[3]),
                           tok[4] == "+", float(tok[9]), ops)
    qspan = sum(o.length for o in ops if o.type in (0, 1))
    tspan = sum(o.length for o in ops if o.type in (0, 2))
    if qspan != abs(pA.end2 - pA.start2) or tspan != abs(pA.end1 - pA.start1):
        raise RuntimeError("Cigar operations do not span the stated coordinates: %r" % line)
    return pA


def cigarRead(fileHandleOrFile):
    fh = open(fileHandleOrFile) if isinstance(fileHandleOrFile, str) else fileHandleOrFile
    try:
        for line in fh:
            if line.startswith("cigar:"):
                yield cigarReadFromString(line)
    finally:
        if isinstance(fileHandleOrFile, str):
            fh.close()


def cigarToString(pA):
    ops = " ".join("%s %i" % (PairwiseAlignment._LETTER[o.type], o.length) for o in pA.operationList)
    return "cigar: %s %i %i %s %s %i %i %s %s %s" % (
        pA.contig2, pA.start2, pA.end2, "+" if pA.strand2 else "-", pA.contig1, pA.start1, pA.end1,
        "+" if pA.strand1 else "-", repr(float(pA.score)) if pA.score != int(pA.score) else "%i" % pA.score, ops)


def cigarWrite(fileHandle, pA, withProbs=False):
    fileHandle.write(cigarToString(pA).rstrip() + "\n")


class Target(object):
    """The slice of jobTree's Target the realign path touches (temp dirs, logging, child / follow-on
    scheduling).  jobTree is an empty submodule in the snapshot and its engine is out of scope (SURVEY.md 2
    row 11): children run immediately and in order, follow-ons right after -- the ordering semantics the
    reference relies on at nanopore/analyses/utils.py:540-609."""

    def __init__(self, tempDir=None):
        self._own = tempDir is None
        self._dir = tempDir or tempfile.mkdtemp(prefix="nanopore_amd_")
        self._followOn = None
        self.messages = []

    def getGlobalTempDir(self):
        return self._dir

    def getLocalTempDir(self):
        return self._dir

    def logToMaster(self, message):
        self.messages.append(message)

    def addChildTargetFn(self, fn, args=()):
        fn(self, *args)

    def addChildTarget(self, target):
        target.run()

    def setFollowOnTargetFn(self, fn, args=()):
        self._followOn = (fn, args)

    def runFollowOns(self):
        while self._followOn is not None:
            fn, args = self._followOn
            self._followOn = None
            fn(self, *args)

    def cleanup(self):
        if self._own and os.path.isdir(self._dir):
            shutil.rmtree(self._dir, ignore_errors=True)
