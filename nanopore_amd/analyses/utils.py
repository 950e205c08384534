"""Host-side mirror of the part of nanopore/analyses/utils.py that sits on the realign path.

Same function names, argument meaning and error behaviour as the reference, Python 3, no pysam / jobTree /
sonLib (all absent from the snapshot).  What changes is the engine: where the reference fans out one jobTree
job and one `cactus_realign` process per SAM record (utils.py:565-570, :576-589) and gathers temp cigar files
(:591-609), `realignSamFile` sends every record of the SAM file to the GPU in ONE batched call through the C
ABI (include/nprealign.h) and splices the returned cigars back in file order.

Reference lines are cited per function.  There is no CPU fallback: without libnprealign.so and a gfx950
device `realignSamFile` raises.
"""
import os
import sys

import numpy as np

from .. import bioio
from .. import sam as pysam  # attribute-compatible subset
from ..bioio import (PairwiseAlignment, cigarRead, cigarReadFromString, fastaRead, fastaWrite, fastqRead,
                     reverseComplement)
from ..hmm import Hmm, SYMBOL_NUMBER
from .hmm_math import (fromMatrix, modifyHmmEmissionsByExpectedVariationRate,  # noqa: F401  (re-exported)
                       normaliseHmmByReferenceGCContent, setHmmIndelEmissionsToBeFlat, toMatrix)

# cactus_realign options hard-coded in the reference's call strings
REALIGN_DIAGONAL_EXPANSION = 10          # utils.py:587
REALIGN_SPLIT_MATRIX_BIGGER_THAN = 3000  # utils.py:587
ANALYSIS_SPLIT_MATRIX_BIGGER_THAN = 100  # alignmentUncertainty.py:41, marginAlignSnpCaller.py:136
CONSTRAINT_DIAGONAL_TRIM = 14            # cactus_realign default, never overridden by the reference


def pathToBaseNanoporeDir():
    """Directory that holds the package (utils.py:76-79): model files live in <base>/nanopore_amd/mappers."""
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.dirname(os.path.dirname(here))


def trainedModelPath(trainedModelFile):
    """nanopore/mappers/<file> of the reference (abstractMapper.py:35)."""
    return os.path.join(pathToBaseNanoporeDir(), "nanopore_amd", "mappers", trainedModelFile)


def getFastaDictionary(fastaFile):
    """First word of each FASTA header -> sequence; names must be unique (utils.py:233-238)."""
    out = {}
    for name, seq in fastaRead(fastaFile):
        key = name.split()[0]
        assert key not in out, "Duplicate fasta sequence name %s" % key
        out[key] = seq
    return out


def getFastqDictionary(fastqFile):
    """First word of each FASTQ header -> sequence (utils.py:240-245)."""
    out = {}
    for name, seq, _ in fastqRead(fastqFile):
        key = name.split()[0]
        assert key not in out, "Duplicate fastq sequence name %s" % key
        out[key] = seq
    return out


def samIterator(sam):
    """Records that have a reference (utils.py:287-293)."""
    for aR in sam:
        if aR.rname != -1:
            yield aR


def clipLengths(alignedRead):
    """(bases clipped before, bases clipped after) the aligned part of SEQ: hard + soft clips at either end of the cigar."""
    cigar = alignedRead.cigar or []
    lead = trail = 0
    for op, length in cigar:
        if op not in (4, 5):
            break
        lead += length
    for op, length in reversed(cigar):
        if op not in (4, 5):
            break
        trail += length
    return lead, trail


def getAbsoluteReadOffset(alignedRead, refSeq, readSeq):
    """Where the first non-clipped base of the record sits in the FASTQ read (utils.py:157-166): the clipped prefix
    (hard clip, then qstart for the soft clip) on the forward strand; on the reverse strand SEQ runs backwards through
    the read, so the offset is counted from the read's last base and comes out non-positive."""
    hard = alignedRead.cigar[0][1] if alignedRead.cigar and alignedRead.cigar[0][0] == 5 else 0
    if not alignedRead.is_reverse:
        return hard + alignedRead.qstart
    return hard + alignedRead.qstart - (len(readSeq) - 1)


def alignedPairs(alignedRead, refLength=None):
    """(reference position, position in alignedRead.query) of every aligned pair of a SAM record, in order.  Pairs whose
    reference position lies beyond the reference are dropped, as the reference drops them (utils.py:145-147: an
    off-by-one of some mappers)."""
    out = []
    for q, r in alignedRead.aligned_pairs:
        if q is None or r is None:
            continue
        if refLength is not None and r >= refLength:
            continue
        out.append((r, q))
    return out


_EXONERATE_OP = {0: "M", 1: "I", 2: "D"}


def getExonerateCigarFormatString(alignedRead, sam):
    """The line cactus_realign reads on stdin (utils.py:168-180):
        cigar: <read> 0 <aligned read bases> + <reference> <pos> <aend> + 1 <op len>...
    read first, reference second, both '+' (a reverse-strand SEQ is already reverse-complemented); clips carry no
    operation; the score field is the literal 1.  The line is parsed back and its match columns counted against the
    record's aligned pairs, the check the reference makes at :179."""
    words = []
    columns = 0
    for op, length in alignedRead.cigar:
        if op in (4, 5):
            continue
        if op not in _EXONERATE_OP:
            raise AssertionError("cigar operation %s of %s has no exonerate form" % (op, alignedRead.qname))
        words.append("%s %i" % (_EXONERATE_OP[op], length))
        columns += length if op == 0 else 0
    line = "cigar: %s 0 %i + %s %i %i + 1 %s" % (alignedRead.qname, alignedRead.qend - alignedRead.qstart,
                                                   sam.getrname(alignedRead.rname), alignedRead.pos, alignedRead.aend, " ".join(words))
    parsed = cigarReadFromString(line)
    assert sum(o.length for o in parsed.operationList if o.type == PairwiseAlignment.PAIRWISE_MATCH) == columns == len(alignedPairs(alignedRead))
    return line


# ---------------------------------------------------------------------------------------------------------
# chaining: local hits -> one global alignment per (read, reference)
# ---------------------------------------------------------------------------------------------------------

def _blockCoordinates(aR, refSeq, readSeq):
    """(first ref pos, first signed read pos, last ref pos, last signed read pos, #aligned pairs)."""
    offset = getAbsoluteReadOffset(aR, refSeq, readSeq)
    pairs = [(q, r) for r, q in alignedPairs(aR, len(refSeq))]
    sign = -1 if aR.is_reverse else 1
    signed = lambda q: sign * abs(offset + q)  # noqa: E731
    return pairs[0][1], signed(pairs[0][0]), pairs[-1][1], signed(pairs[-1][0]), len(pairs)


def chainFn(alignedReads, refSeq, readSeq, scoreFn=None, maxGap=200):
    """Highest-scoring co-linear chain of local alignments on one strand; score = number of aligned pairs; two blocks
    chain when the second starts after the first ends in both sequences and the summed gap is at most maxGap
    (utils.py:388-426).  The scan itself is native (include/nprealign.h: npr_chain_hits -- sort + windowed scan instead of
    the reference's all-pairs loop, same chain, ties included)."""
    import numpy as np
    from .. import _lib
    alignedReads = list(alignedReads)
    coords = [_blockCoordinates(aR, refSeq, readSeq) for aR in alignedReads]
    n = len(alignedReads)
    col = lambda k: np.ascontiguousarray([c[k] for c in coords], dtype=np.int64)  # noqa: E731
    score = np.ascontiguousarray([scoreFn(aR, refSeq, readSeq) for aR in alignedReads] if scoreFn else [c[4] for c in coords], dtype=np.int64)
    reverse = np.ascontiguousarray([1 if aR.is_reverse else 0 for aR in alignedReads], dtype=np.uint8)
    chain = np.zeros(max(n, 1), dtype=np.int64)
    k = _lib.load().npr_chain_hits(n, _lib.ptr(col(0)), _lib.ptr(col(1)), _lib.ptr(col(2)), _lib.ptr(col(3)), _lib.ptr(reverse), _lib.ptr(score),
                                   maxGap, _lib.ptr(chain))
    if k < 0:
        raise _lib.NprError(int(k), "npr_chain_hits")
    return [alignedReads[int(i)] for i in chain[:k]]


def mergeChainedAlignedReads(chainedAlignedReads, refSequence, readSequence):
    """The chain's blocks as ONE global record: pos 0, SEQ = the whole read in the strand's orientation, cigar over the entire
    reference and the entire read (utils.py:295-386; spans asserted at :381-382).  The cigar is made natively from the
    blocks' coordinates (include/nprealign.h: npr_chain_merge): what lies between, before and after the blocks becomes
    explicit D / I operations; blocks out of chain order or overlapping are an AssertionError, as in the reference."""
    from .. import _lib
    blocks = list(chainedAlignedReads)
    head = blocks[0]
    if any(aR.is_reverse != head.is_reverse for aR in blocks):
        raise AssertionError("blocks of one chain lie on both strands (%s)" % head.qname)
    for aR in blocks:  # the reference asserts `op in (0, 1, 2, 4, 5)` (utils.py:357, :366): an = / X / N / P block would be under-counted
        for op, _ in aR.cigar:
            if op not in (0, 1, 2, 4, 5):
                raise AssertionError("cigar operation %d of %s is outside M I D S H" % (op, aR.qname))
    guides = [_guideOf(aR) for aR in blocks]
    ref_pos = np.array([aR.pos for aR in blocks], dtype=np.int64)
    read_pos = np.array([clipLengths(aR)[0] for aR in blocks], dtype=np.int64)  # first aligned base in SEQ orientation, either strand
    ops_off = np.zeros(len(blocks) + 1, dtype=np.int64)
    np.cumsum([len(g) for g in guides], out=ops_off[1:])
    ops = np.array([o for g in guides for o in g], dtype=np.int32).reshape(-1, 2)
    out = np.zeros((int(ops_off[-1]) + 2 * len(blocks) + 2, 2), dtype=np.int32)
    k = _lib.load().npr_chain_merge(len(blocks), _lib.ptr(ref_pos), _lib.ptr(read_pos), _lib.ptr(ops_off), _lib.ptr(ops), len(refSequence),
                                    len(readSequence), _lib.ptr(out), len(out))
    if k == _lib.ERR_INVALID:
        raise AssertionError("chain of %s is not co-linear inside its sequences" % head.qname)
    if k < 0:
        raise _lib.NprError(int(k), "npr_chain_merge")
    merged = pysam.AlignedRead()
    merged.qname, merged.rname, merged.pos = head.qname, head.rname, 0
    merged.flag = 0x10 if head.is_reverse else 0
    merged.seq = reverseComplement(readSequence) if head.is_reverse else readSequence
    merged.cigar = [(int(op), int(n)) for op, n in out[:k]]
    return merged


def chainSamFile(samFile, outputSamFile, readFastqFile, referenceFastaFile, chainFn=chainFn):
    """At most one global alignment per (read, reference) (utils.py:441-469)."""
    sam = pysam.Samfile(samFile, "r")
    refSequences = getFastaDictionary(referenceFastaFile)
    readSequences = getFastqDictionary(readFastqFile)
    buckets = {}
    for aR in samIterator(sam):
        if aR.qname not in readSequences:
            raise RuntimeError("Aligned read name: %s not in read sequences names" % aR.qname)
        buckets.setdefault((aR.qname, aR.rname), []).append(aR)
    out = pysam.Samfile(outputSamFile, "wh", template=sam)
    chained = []
    for (readName, refID), alignedReads in buckets.items():
        refSeq = refSequences[sam.getrname(refID)]
        readSeq = readSequences[readName]
        chained.append(mergeChainedAlignedReads(chainFn(alignedReads, refSeq, readSeq), refSeq, readSeq))
    chained.sort(key=lambda a: (a.rname, a.pos, a.qname))
    for cAR in chained:
        out.write(cAR)
    sam.close()
    out.close()


# ---------------------------------------------------------------------------------------------------------
# realign (the hot path)
# ---------------------------------------------------------------------------------------------------------

_shared_ctx = {}


def _context(device=0):
    from .. import realign
    if device not in _shared_ctx:
        _shared_ctx[device] = realign.Context(device)
    return _shared_ctx[device]


def _loadHmmInto(ctx, hmmFile, slot=0):
    ctx.set_hmm(None if hmmFile is None else Hmm.loadHmm(hmmFile), slot=slot)


def _guideOf(aR):
    return [(op, length) for op, length in aR.cigar if op in (0, 1, 2)]


def realignRecords(sam, records, refSequences, gapGamma, matchGamma, hmmFile, mode=None, splitThreshold=None,
                   ctx=None, want_pairs=False):
    """All records to the GPU in one batched call.  Returns the list of per-record result dicts in input
    order.  Replaces the child-per-record fan-out of utils.py:565-570 and the cactus_realign call :587."""
    from .. import realign
    ctx = ctx or _context()
    _loadHmmInto(ctx, hmmFile)
    names = sorted(refSequences)
    index = {n: i for i, n in enumerate(names)}
    params = realign.make_params(
        band_mode=realign.BAND_ANCHOR, diagonal_expansion=REALIGN_DIAGONAL_EXPANSION,
        constraint_trim=CONSTRAINT_DIAGONAL_TRIM,
        split_threshold=REALIGN_SPLIT_MATRIX_BIGGER_THAN if splitThreshold is None else splitThreshold,
        gap_gamma=gapGamma, match_gamma=matchGamma, mode=realign.MODE_REALIGN if mode is None else mode)
    reads, guides, ref_index, starts = [], [], [], []
    for aR in records:
        for op, _ in aR.cigar:
            assert op in (0, 1, 2, 4, 5)
        reads.append(aR.query)
        guides.append(_guideOf(aR))
        ref_index.append(index[sam.getrname(aR.rname)])
        # the window the exonerate cigar names: reference [pos, aend), read [0, len(query)) (utils.py:175-177).  A chained
        # record is global (pos 0, aend = reference length, utils.py:381-382); a local hit of an un-chained mapper SAM
        # (AlignmentUncertainty on a base mapper's output) is realigned inside its own window, as cactus_realign does.
        starts.append((aR.pos, 0))
    refs = [refSequences[n] for n in names]

    def run(lo, hi):
        # one batched call; a batch the device cannot hold (NPR_ERR_NOMEM: the reference's per-read jobs have no such
        # limit) is halved and retried, results concatenated in input order
        try:
            return ctx.realign(params, refs, reads[lo:hi], guides[lo:hi], ref_index=ref_index[lo:hi], want_pairs=want_pairs,
                               guide_start=starts[lo:hi])
        except realign.NprError as e:
            if e.code != realign.ERR_NOMEM or hi - lo < 2:
                raise
        mid = (lo + hi) // 2
        return run(lo, mid) + run(mid, hi)

    return run(0, len(records))


def realignSamFile(samFile, outputSamFile, readFastqFile, referenceFastaFile, hmmFile, gapGamma, matchGamma, ctx=None, group=None):
    """realignSamFile2TargetFn + realignCigarTargetFn + realignSamFile3TargetFn (utils.py:557-609): output SAM == input SAM
    with only each record's CIGAR replaced, same order, header copied (utils.py:596-605).

    The whole file goes through `nanopore_amd.job.realign_sam_file`: the text is mapped and parsed natively (no Python
    object per record), the records shard over the ranks of an initialised torch.distributed process group (collective
    call; one rank otherwise), every rank keeps two batches in flight on its GPU and writes its own block of the output.
    `readFastqFile` is accepted for the reference's signature: like realignSamFile2TargetFn, which loads the FASTQ and then
    hands cactus_realign aR.query (utils.py:561, :570), the realignment reads its sequences from the SAM records.
    Returns the per-record results (structured array, input order) on rank 0, None on the other ranks."""
    from .. import job
    if ctx is not None:  # a caller's context joins the pool of its GPU, so its models / scratch are reused
        pool = job._ctx_pool.setdefault(ctx.device, [])
        if ctx not in pool:
            pool.insert(0, ctx)
    out = job.realign_sam_file(samFile, outputSamFile, referenceFastaFile, hmm=hmmFile, gapGamma=gapGamma, matchGamma=matchGamma,
                               group=group, gpu=None if ctx is None else ctx.device)
    if "results" not in out:
        return None
    results = out["results"]
    failed = np.nonzero(results["status"] != 0)[0]
    if len(failed):
        # the reference's system() raises on a non-zero exit of cactus_realign and the job tree reports failed jobs
        # (pipeline.py:209-210), leaving no output SAM; a batch finds out after the fact
        if os.path.exists(outputSamFile):
            os.unlink(outputSamFile)
        raise RuntimeError("Realignment failed for %d record(s), first: record %d, status %d" % (len(failed), int(failed[0]), int(results["status"][failed[0]])))
    return results


def realignSamFileByRecord(samFile, outputSamFile, readFastqFile, referenceFastaFile, hmmFile, gapGamma, matchGamma, ctx=None):
    """The same through the record-at-a-time host mirror (Samfile iterator, one AlignedRead per record, realignRecords, one
    write per record): the shape of the reference's own loop, kept as the small-input path of the analyses that hold
    AlignedRead objects anyway and as the cross-check of the bulk path (tests/test_gpu_job.py: byte-identical output)."""
    refSequences = getFastaDictionary(referenceFastaFile)
    sam = pysam.Samfile(samFile, "r")
    records = list(samIterator(sam))
    results = realignRecords(sam, records, refSequences, gapGamma, matchGamma, hmmFile, ctx=ctx)
    failed = [(aR.qname, r["status"]) for aR, r in zip(records, results) if r["status"] != 0]
    if failed:
        raise RuntimeError("Realignment failed for %d record(s), first: %s" % (len(failed), failed[0]))
    out = pysam.Samfile(outputSamFile, "wh", template=sam)
    for aR, r in zip(records, results):
        assert len(r["ops"]) > 0 or len(aR.query) == 0          # exactly one cigar per record (utils.py:588-589)
        aR.cigar = [(op, length) for op, length in r["ops"]]    # utils.py:602
        out.write(aR)
    sam.close()
    out.close()
    return results


EM_SPLIT_MATRIX_BIGGER_THAN = 300  # options.optionsToRealign, utils.py:511


def stageSamFile(samFile, referenceFastaFile, splitThreshold, gapGamma=0.5, matchGamma=0.0, mode=None, ctx=None, maxPairsPerBase=0):
    """Stages every record of a chained SAM file on the GPU (band planning + upload); returns (batch, sam, records)."""
    from .. import realign
    ctx = ctx or _context()
    refSequences = getFastaDictionary(referenceFastaFile)
    sam = pysam.Samfile(samFile, "r")
    records = list(samIterator(sam))
    names = sorted(refSequences)
    index = {n: i for i, n in enumerate(names)}
    params = realign.make_params(band_mode=realign.BAND_ANCHOR, diagonal_expansion=REALIGN_DIAGONAL_EXPANSION,
                                 constraint_trim=CONSTRAINT_DIAGONAL_TRIM, split_threshold=splitThreshold,
                                 gap_gamma=gapGamma, match_gamma=matchGamma,
                                 mode=realign.MODE_REALIGN if mode is None else mode, max_pairs_per_base=maxPairsPerBase)
    batch = ctx.stage(params, [refSequences[n] for n in names], [aR.query for aR in records], [_guideOf(aR) for aR in records],
                      ref_index=[index[sam.getrname(aR.rname)] for aR in records], guide_start=[(aR.pos, 0) for aR in records])
    return batch, sam, records


def stageSamFileForTraining(samFile, readFastqFile, referenceFastaFile, options, ctx=None):
    """The trainer's view of a chained SAM file, without a Python object per record: the file is mapped and parsed by the native
    scanners (nanopore_amd/ingest.py), the global-alignment shape the trainer relies on is asserted on the parsed fields of EVERY
    record, sequences included (utils.py:492-501), the alignments are sampled
    and cut into batches as options.maxAlignmentLengthToSample / maxAlignmentLengthPerJob say (em.sampleAlignments) and each batch
    is staged once, in NPR_MODE_EXPECTATIONS, for all iterations of all trials.  Returns an em.BatchSet (alignments, taken)."""
    from .. import em, ingest, job, realign
    ctx = ctx or _context()
    sam = ingest.SamText(samFile)
    fasta = ingest.FastaTable(referenceFastaFile)
    fields = sam.parse()
    keep = sam.records_with_a_reference(fields, sam.span)
    fields, span = fields[keep], sam.span[keep]
    n = len(fields)
    if n == 0:
        raise RuntimeError("No alignments to train on in %s" % samFile)
    qnames, qtext, qspan = ingest.fastq_table(readFastqFile)
    qlen = dict(zip(qnames, (qspan[:, 1] - qspan[:, 0]).tolist()))
    qat = dict(zip(qnames, range(len(qnames))))
    ref_len = np.array([fasta.off[fasta.index[name] + 1] - fasta.off[fasta.index[name]] for name in sam.references] + [0], dtype=np.int64)
    F = ingest
    names = [sam.field_bytes(int(a), int(b)).decode() for a, b in zip(span[:, 0], fields[:, F.F_QNAME_END])]
    read_len = np.array([qlen[q] for q in names], dtype=np.int64)  # (KeyError: a record of a read the FASTQ does not have)
    assert (fields[:, F.F_POS] == 0).all()                                                     # aR.pos == 0
    assert (fields[:, F.F_QUERY_LO] == fields[:, F.F_SEQ_LO]).all()                            # aR.qstart == 0
    assert (fields[:, F.F_QUERY_HI] - fields[:, F.F_QUERY_LO] == read_len).all()               # aR.qend == len(read)
    assert (fields[:, F.F_REF_SPAN] == ref_len[fields[:, F.F_TID]]).all()                      # aR.aend == len(reference)
    # aR.query == the read / its reverse complement, for EVERY record as the reference asserts it (utils.py:496-501): byte tables for
    # the case fold and the complement, one vector compare per record on the mapped texts (no strings made)
    up = np.arange(256, dtype=np.uint8)
    up[ord("a"):ord("z") + 1] -= 32
    rc = up.copy()
    for x, y in zip(b"ACGTN", b"TGCAN"):
        rc[x] = rc[x + 32] = y
    sam_text, fq_text = np.asarray(sam.text), np.asarray(qtext)
    for i in range(n):
        a, b = qspan[qat[names[i]]]
        q = up[sam_text[fields[i, F.F_QUERY_LO]:fields[i, F.F_QUERY_HI]]]
        read = rc[fq_text[a:b]][::-1] if fields[i, F.F_FLAG] & 0x10 else up[fq_text[a:b]]
        assert np.array_equal(q, read), "record %d (%s): SEQ is neither the read nor its reverse complement" % (i, names[i])
    src = job.SamSource(sam, fasta, span, fields)
    cols = np.zeros(n, dtype=np.int64)                                                       # alignment columns of every record
    np.add.at(cols, np.repeat(np.arange(n), src.guide_off[1:] - src.guide_off[:-1]), src.guide_ops[:, 1])
    parts = em.sampleAlignments(cols, options, rng=np.random.default_rng(options.seed))
    params = realign.make_params(band_mode=realign.BAND_ANCHOR, diagonal_expansion=REALIGN_DIAGONAL_EXPANSION,
                                 constraint_trim=CONSTRAINT_DIAGONAL_TRIM, split_threshold=EM_SPLIT_MATRIX_BIGGER_THAN,
                                 mode=realign.MODE_EXPECTATIONS)
    batches = []
    try:
        for idx in parts:
            batches.append(src.stage_records(ctx, params, idx))
    except BaseException:
        for b in batches:
            b.close()
        raise
    bs = em.BatchSet(batches)
    bs.alignments, bs.taken = n, int(sum(len(p) for p in parts))
    return bs


def learnModelFromSamFileTargetFn(target, samFile, readFastqFile, referenceFastaFile, outputModel, options=None, ctx=None):
    """Does expectation maximisation on a (chained) sam file to learn the hmm for it (utils.py:471-531).  The
    unnormalised model goes to <outputModel>_unnormalised (skipped if it exists, :527), the XML summary to
    <outputModel>.xml (:518), and learnModelFromSamFileTargetFn2 normalises it into <outputModel>."""
    from .. import em
    if options is None:
        options = em.Options()
    options.outputXMLModelFile = outputModel + ".xml"
    unnormalisedOutputModel = outputModel + "_unnormalised"
    if not os.path.exists(unnormalisedOutputModel):
        # (the asserts on the records' global-alignment shape, utils.py:492-501, are made on the natively parsed fields)
        batch = stageSamFileForTraining(samFile, readFastqFile, referenceFastaFile, options, ctx=ctx)
        try:
            if target is not None and batch.taken < batch.alignments:
                target.logToMaster("EM: %d of %d alignments sampled (maxAlignmentLengthToSample %d), %d batch(es)"
                                   % (batch.taken, batch.alignments, options.maxAlignmentLengthToSample, len(batch.batches)))
            em.expectationMaximisationTrials(batch, unnormalisedOutputModel, options,
                                             log=(target.logToMaster if target is not None else None))
        finally:
            batch.close()
    learnModelFromSamFileTargetFn2(target, unnormalisedOutputModel, outputModel)


def learnModelFromSamFileTargetFn2(target, unnormalisedOutputModel, outputModel):
    """Flat indel emissions, reference base frequencies normalised to GC 0.5 (utils.py:533-538)."""
    hmm = Hmm.loadHmm(unnormalisedOutputModel)
    setHmmIndelEmissionsToBeFlat(hmm)
    normaliseHmmByReferenceGCContent(hmm, 0.5)
    hmm.write(outputModel)


def realignSamFileTargetFn(target, samFile, outputSamFile, readFastqFile, referenceFastaFile, gapGamma, matchGamma,
                           hmmFile=None, trainHmmFile=False, chainFn=chainFn):
    """Chains and then realigns the resulting global alignments (utils.py:540-555).  `target` only supplies the
    temp directory, as in the reference; pass a nanopore_amd.bioio.Target (or None for a private one)."""
    own = target is None
    target = target or bioio.Target()
    try:
        tempSamFile = os.path.join(target.getGlobalTempDir(), "temp.sam")
        chainSamFile(samFile, tempSamFile, readFastqFile, referenceFastaFile, chainFn)
        if hmmFile is not None and trainHmmFile:
            learnModelFromSamFileTargetFn(target, tempSamFile, readFastqFile, referenceFastaFile, hmmFile)
        else:
            assert not trainHmmFile
        return realignSamFile(tempSamFile, outputSamFile, readFastqFile, referenceFastaFile, hmmFile, gapGamma, matchGamma)
    finally:
        if own:
            target.cleanup()


def writePosteriorProbs(path, x, y, p):
    """`refPosition readPosition posteriorProb` per line: the --outputAllPosteriorProbs / --outputPosteriorProbs TSV
    parsed at marginAlignSnpCaller.py:149."""
    with open(path, "w") as fh:
        for a, b, c in zip(x, y, p):
            fh.write("%i\t%i\t%s\n" % (a, b, repr(float(c))))
