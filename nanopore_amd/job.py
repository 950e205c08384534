"""The whole realignment job on one read set: files (or resident arrays) in, realigned SAM out, sharded over the ranks of a
node and pipelined on every rank.

The reference's job is: one jobTree job per record of ONE SAM file (nanopore/analyses/utils.py:565-570), the temp cigar
files gathered in input order and spliced into a copy of the input SAM (utils.py:591-609).  Here:

* every rank takes a CONTIGUOUS range of the records, balanced by record length (`dist.shard_ranges`), so the output needs no
  data-path collective: a rank writes the records of its range at its own offset of the one output file (an all_gather of
  the block sizes gives the offsets); one small gather (RCCL under backend nccl) brings the per-read results to rank 0 for
  the summary;
* a rank cuts its range into chunks of about `CHUNK_BASES` read bases and runs them as a pipeline with `WORKERS` (3) in
  flight, one thread per phase: stage (plan + pack + H2D + device planner) -> DP -> finish (MEA chain + cigar on the
  device) -> fetch + splice / format of the chunk's SAM records -> the caller's thread writes the blocks in order.  Chunk k
  lives in realigner context k mod 3 (stream, staging buffers) on the rank's GPU, so the stager runs up to two chunks ahead
  of the DP and the host phases of one chunk run under the DP sweep of another (ctypes releases the GIL inside the C
  ABI; the contexts of a device share its forward scratch, whose mutex lets one DP pass or one MEA stage run at a time
  -- each fills the chip anyway).

`realign_sam_file` is what `analyses.utils.realignSamFile` (the plugin surface: AbstractMapper.realignSamFile,
realignSamFileTargetFn) runs; `run_job` is the same pipeline over resident synthetic arrays (`bench.py --workload c3`).
"""
import gc
import os
import queue
import threading
import time
import xml.etree.ElementTree as ET

import numpy as np

from . import _lib
from . import dist as npd

_DEFAULT_CHUNK_BASES = 50_000_000
CHUNK_BASES = int(os.environ.get("NPR_JOB_CHUNK_BASES", _DEFAULT_CHUNK_BASES))  # ~6 k reads of 8 kb (round 5; rounds 3-4: 10^8, see chunk_bounds)
MIN_CHUNK_READS = int(os.environ.get("NPR_JOB_MIN_CHUNK_READS", 4096))  # reads of a chunk at the default chunk size (a caller who asks for smaller chunks gets them)
WORKERS = int(os.environ.get("NPR_JOB_WORKERS", 3))  # chunks in flight (contexts per GPU): one in its DP, one being finished / fetched, one staged ahead
TRACE = os.environ.get("NPR_JOB_TRACE") is not None  # timings["trace"]: (phase, start, end) per chunk, seconds (tools/job_trace.py)


# ---------------------------------------------------------------------------------------------------------
# record sources: what a set of reads looks like to the pipeline
# ---------------------------------------------------------------------------------------------------------

class ArraySource(object):
    """Reads, guides and references as flat arrays (the C ABI's batch arguments), plus a formatter for the output records
    of a range.  read i = text[read_begin[i] : read_end[i]]."""

    def __init__(self, ref, ref_off, text, read_begin, read_end, guide_ops, guide_off, ref_index=None, guide_start=None,
                 model_slot=None):
        self.ref = np.ascontiguousarray(ref, dtype=np.uint8)
        self.ref_off = np.ascontiguousarray(ref_off, dtype=np.int64)
        self.text = text
        self.read_begin = np.ascontiguousarray(read_begin, dtype=np.int64)
        self.read_end = np.ascontiguousarray(read_end, dtype=np.int64)
        self.guide_ops = None if guide_ops is None else np.ascontiguousarray(guide_ops, dtype=np.int32).reshape(-1, 2)  # (None: the subclass has its own)
        self.guide_off = None if guide_off is None else np.ascontiguousarray(guide_off, dtype=np.int64)
        self.ref_index = None if ref_index is None else np.ascontiguousarray(ref_index, dtype=np.int32)
        self.guide_start = None if guide_start is None else np.ascontiguousarray(guide_start, dtype=np.int64).reshape(-1, 2)
        self.model_slot = None if model_slot is None else np.ascontiguousarray(model_slot, dtype=np.int32)
        self.n = len(self.read_begin)
        if self.ref_index is None and len(self.ref_off) - 1 != self.n:
            raise ValueError("without ref_index there must be one reference slice per read")

    def lengths(self):
        return self.read_end - self.read_begin

    def stage(self, ctx, params, lo, hi):
        sl = slice(lo, hi)
        if self.ref_index is None:  # per-read slices: the range's own part of the reference table
            ref, ref_off, ri = self.ref, self.ref_off[lo:hi + 1], None
        else:
            ref, ref_off, ri = self.ref, self.ref_off, self.ref_index[sl]
        return ctx.stage_spans(params, ref, ref_off, self.text, self.read_begin[sl], self.read_end[sl], self.guide_ops,
                               self.guide_off[lo:hi + 1], model_slot=None if self.model_slot is None else self.model_slot[sl],
                               ref_index=ri, guide_start=None if self.guide_start is None else self.guide_start[sl])

    def stage_records(self, ctx, params, idx):
        """The records `idx` (any subset, in that order) as one batch: the few reads of a chunk that have to run again."""
        idx = np.asarray(idx, dtype=np.int64)
        k = self.guide_off[idx + 1] - self.guide_off[idx]
        goff = np.zeros(len(idx) + 1, dtype=np.int64)
        np.cumsum(k, out=goff[1:])
        gops = self.guide_ops[np.repeat(self.guide_off[idx] - goff[:-1], k) + np.arange(int(goff[-1]))]
        if self.ref_index is None:  # per-read slices: a private table of the subset's slices, read where they lie
            ri = np.arange(len(idx), dtype=np.int32)
            lens = self.ref_off[idx + 1] - self.ref_off[idx]
            ref_off = np.zeros(len(idx) + 1, dtype=np.int64)
            np.cumsum(lens, out=ref_off[1:])
            ref = self.ref[np.repeat(self.ref_off[idx] - ref_off[:-1], lens) + np.arange(int(ref_off[-1]))]
        else:
            ref, ref_off, ri = self.ref, self.ref_off, self.ref_index[idx]
        return ctx.stage_spans(params, ref, ref_off, self.text, self.read_begin[idx], self.read_end[idx], gops, goff,
                               model_slot=None if self.model_slot is None else self.model_slot[idx], ref_index=ri,
                               guide_start=None if self.guide_start is None else self.guide_start[idx])

    def format_block(self, lo, hi, ops_off, words):
        raise NotImplementedError


class SynthSource(ArraySource):
    """A synthetic workload dict (nanopore_amd.synth): records are made up from the arrays (QNAME read_<i>, POS = where the
    guide's window starts), formatted natively (npr_format_sam_records)."""

    def __init__(self, w, model_slot=None, ref_names=None):
        ro = np.asarray(w["read_off"], dtype=np.int64)
        ArraySource.__init__(self, w["ref"], w["ref_off"], np.ascontiguousarray(w["read"], dtype=np.uint8), ro[:-1], ro[1:], w["guide_ops"],
                             w["guide_off"], ref_index=w.get("ref_index"), guide_start=w.get("guide_start"), model_slot=model_slot)
        n_refs = len(self.ref_off) - 1
        self.ref_names = ref_names or ["ref_%d" % k for k in range(n_refs)]
        head = [b"@HD\tVN:1.0\tSO:unsorted\n"]
        for k in range(n_refs):
            head.append(("@SQ\tSN:%s\tLN:%d\n" % (self.ref_names[k], int(self.ref_off[k + 1] - self.ref_off[k]))).encode())
        self.header = b"".join(head)

    def format_block(self, lo, hi, ops_off, words):
        from . import realign
        n = hi - lo
        nops = np.asarray(ops_off[1:]) - np.asarray(ops_off[:-1])
        ref_index = self.ref_index[lo:hi] if self.ref_index is not None else np.arange(lo, hi, dtype=np.int32)
        pos = (self.guide_start[lo:hi, 0] if self.guide_start is not None else np.zeros(n, dtype=np.int64)) + 1
        qnames = [b"read_%d" % i for i in range(lo, hi)]
        a = int(self.read_begin[lo]) if n else 0
        b = int(self.read_end[hi - 1]) if n else 0
        seq_off = np.concatenate([self.read_begin[lo:hi] - a, [b - a]]).astype(np.int64)
        buf, _ = realign.format_sam_records(qnames, [s.encode() for s in self.ref_names], ref_index, pos, ops_off[:-1], nops, words,
                                            self.text[a:b], seq_off)
        return buf


class SamSource(ArraySource):
    """The records of a mapped SAM text (nanopore_amd.ingest.SamText) against a FASTA table: the realigner sees aR.query
    (the aligned part of SEQ, utils.py:570) where it lies in the file's bytes, the guide = the record's M / I / D operations
    starting at (aR.pos, 0) (the exonerate line's coordinates, utils.py:173-177); an output record is the input record with
    its CIGAR field replaced (utils.py:597-605)."""

    def __init__(self, sam, fasta, span, fields, model_slot=None):
        from . import ingest as ing
        self.sam = sam
        self.span = np.ascontiguousarray(span, dtype=np.int64)
        self.fields = np.ascontiguousarray(fields, dtype=np.int64)
        self._guides = None  # (guide_off, guide_ops) of ALL records, built when someone asks; the job builds a chunk's as it stages it, in a pooled buffer
        self._F_GUIDE_OPS = ing.F_GUIDE_OPS
        tid_to_ref = np.array([fasta.index.get(name, -1) for name in sam.references] + [-1], dtype=np.int32)
        ref_index = tid_to_ref[self.fields[:, ing.F_TID]]
        if len(ref_index) and (ref_index < 0).any():
            k = int(np.nonzero(ref_index < 0)[0][0])
            raise KeyError(sam.references[int(self.fields[k, ing.F_TID])])  # refSequences[sam.getrname(aR.rname)] (utils.py:570)
        gs = np.zeros((len(self.fields), 2), dtype=np.int64)
        gs[:, 0] = self.fields[:, ing.F_POS]
        ArraySource.__init__(self, fasta.seq, fasta.off, sam.text, self.fields[:, ing.F_QUERY_LO], self.fields[:, ing.F_QUERY_HI], None,
                             None, ref_index=ref_index, guide_start=gs, model_slot=model_slot)
        self.header = sam.header

    # The guides (the M / I / D operations of every cigar) of a 50 000-record file take 25 ms to build, with nothing else running: a
    # chunk's are built when it is staged, under the DP pass of the chunk before (round 4); the whole table only for who reads it.
    def _all_guides(self):
        if self._guides is None:
            self._guides = self.sam.guides(self.fields)
        return self._guides

    guide_off = property(lambda self: self._all_guides()[0], lambda self, value: None)
    guide_ops = property(lambda self: self._all_guides()[1], lambda self, value: None)

    def stage(self, ctx, params, lo, hi):
        # (one stager thread per source; the batch holds its own copy of the guides when stage_spans returns.  The operations of a
        # chunk of 12 500 reads are 300 MB: allocated and released per chunk they cost two rounds of page faults and an munmap under the
        # interpreter lock, during which no other phase of the pipeline can pick up its next chunk)
        sl = slice(lo, hi)
        buf = _take(8 * int(np.sum(self.fields[sl, self._F_GUIDE_OPS])))
        goff = gops = None
        try:
            goff, gops = self.sam.guides(self.fields[sl], buffer=buf.view(np.int32))
            return ctx.stage_spans(params, self.ref, self.ref_off, self.text, self.read_begin[sl], self.read_end[sl], gops, goff,
                                   model_slot=None if self.model_slot is None else self.model_slot[sl], ref_index=self.ref_index[sl],
                                   guide_start=self.guide_start[sl])
        finally:
            del goff, gops
            _give(buf)

    def stage_records(self, ctx, params, idx):
        idx = np.asarray(idx, dtype=np.int64)
        goff, gops = self.sam.guides(self.fields[idx])
        return ctx.stage_spans(params, self.ref, self.ref_off, self.text, self.read_begin[idx], self.read_end[idx], gops, goff,
                               model_slot=None if self.model_slot is None else self.model_slot[idx], ref_index=self.ref_index[idx],
                               guide_start=self.guide_start[idx])

    def format_block(self, lo, hi, ops_off, words):
        nops = np.asarray(ops_off[1:]) - np.asarray(ops_off[:-1])
        return self.sam.splice(self.span[lo:hi], self.fields[lo:hi], ops_off[:-1], nops, words, take=_take)  # (run_pipeline gives it back)


# ---------------------------------------------------------------------------------------------------------
# the per-rank pipeline
# ---------------------------------------------------------------------------------------------------------

_ctx_pool = {}


def contexts(device, count):
    """`count` realigner contexts on one GPU, kept for the life of the process (a context owns a stream and a scratch arena
    whose allocation costs more than a small job)."""
    from . import realign
    pool = _ctx_pool.setdefault(device, [])
    pool[:] = [c for c in pool if getattr(c, "_h", None)]  # (a caller may have closed one it lent)
    while len(pool) < count:
        pool.append(realign.Context(device))
    return pool[:count]


def close_contexts():
    for pool in _ctx_pool.values():
        for c in pool:
            c.close()
    _ctx_pool.clear()
    with _host_lock:
        _host_pool.clear()


# Host buffers of a job's chunks -- the guide operations being staged (300 MB for 12 500 reads), the packed cigars fetched (150 MB),
# the formatted records on their way to the file (2 x 75 MB) -- kept from chunk to chunk and from job to job, like the contexts:
# allocated and released per chunk they are page faults on the way in and an munmap under the interpreter lock on the way out,
# 5-25 ms during which no phase of the pipeline can pick up its next chunk (round 4).  At most _HOST_POOL_MAX buffers are kept;
# close_contexts() drops them.
_HOST_POOL_MAX = 8
_HOST_POOL_BYTES = int(os.environ.get("NPR_JOB_HOST_POOL_MB", 1536)) << 20  # kept between chunks; run_source trims to a quarter when a job ends
_host_pool = []
_host_lock = threading.Lock()


def _trim_host_pool(limit):
    """Drops the largest buffers of the pool until what is kept fits `limit` bytes."""
    with _host_lock:
        _host_pool.sort(key=lambda b: b.nbytes)
        while _host_pool and sum(b.nbytes for b in _host_pool) > limit:
            _host_pool.pop()


def _take(nbytes):
    """A uint8 buffer of at least nbytes (a multiple of 4096 bytes long): the smallest of the pool that fits, else a new one."""
    with _host_lock:
        best = None
        for i, b in enumerate(_host_pool):
            if b.nbytes >= nbytes and (best is None or b.nbytes < _host_pool[best].nbytes):
                best = i
        if best is not None:
            return _host_pool.pop(best)
    return np.empty((nbytes + nbytes // 8 + 4095) // 4096 * 4096, dtype=np.uint8)


def _give(buf):
    """A buffer from _take (or any view of it) back to the pool."""
    while isinstance(buf, np.ndarray) and buf.base is not None:
        buf = buf.base
    if not isinstance(buf, np.ndarray) or buf.dtype != np.uint8 or buf.nbytes % 4096:
        return
    with _host_lock:
        if (len(_host_pool) < _HOST_POOL_MAX and not any(b is buf for b in _host_pool)
                and sum(b.nbytes for b in _host_pool) + buf.nbytes <= _HOST_POOL_BYTES):
            _host_pool.append(buf)


_gc_lock = threading.Lock()
_gc_jobs = [0, False]  # pipelines that hold the collector off; whether it was on when the first of them came


def _gc_hold():
    """The cyclic collector off for the duration of a pipeline; counted, so that jobs running side by side in one process switch it back
    on once, when the last of them ends, and only if it was on."""
    with _gc_lock:
        if _gc_jobs[0] == 0:
            _gc_jobs[1] = gc.isenabled()
            if _gc_jobs[1]:
                gc.disable()
        _gc_jobs[0] += 1
    return True


def _gc_release():
    with _gc_lock:
        _gc_jobs[0] -= 1
        if _gc_jobs[0] == 0 and _gc_jobs[1]:
            gc.enable()


def chunk_bounds(lengths, lo, hi, chunk_bases=None, workers=None):
    """[lo, hi) cut into chunks of about chunk_bases read bases (equal shares of the bases), none below MIN_CHUNK_READS reads at the
    default chunk size."""
    min_reads = MIN_CHUNK_READS if (chunk_bases is None and CHUNK_BASES == _DEFAULT_CHUNK_BASES) else 1
    chunk_bases = chunk_bases or CHUNK_BASES
    workers = workers or WORKERS
    n = hi - lo
    if n <= 0:
        return []
    total = float(np.sum(lengths[lo:hi]))
    k = max(1, int(round(total / chunk_bases)))
    # A DP launch lasts at least as long as its longest read's chain.  On one wavefront per read (rounds 3-4) a 20 kb read took 45 ms, what
    # 12 000 reads of 8 kb take when they fill the chip, so chunks held 12 288 reads or more and 10^8 bases.  With a read's sweeps on two
    # wavefronts that meet in the middle (round 5) the chain is 21 ms and chunks of half the size pay: measured on one GPU, ms per step for
    # 12 500 / 25 000 / 50 000 reads of 8 kb -- chunks of 10^8 bases 143 / 232 / 380, of 5 x 10^7 122 / 198 / 367, of 3.3 x 10^7 115 / 211 / 398,
    # of 2.5 x 10^7 123 / 223 / 431 (6 250 reads as one chunk 75, as two 78).  A range worth two chunks takes three: with two, nothing runs
    # beside the first one's staging or the second one's finishing.
    k = max(1, min(k, n // min_reads))
    if k == 2 and n >= 3 * min_reads:
        k = 3
    # ... but never more than four times the bases asked for: MIN_CHUNK_READS reads of 50-100 kb would be six to twelve chunks' worth, three
    # of them in flight -- such a job would live in the halve-and-stage-again path that is meant for the exception
    k = max(k, int(total // (4 * chunk_bases)))
    k = min(k, n)
    cuts = lo + npd.shard_ranges(lengths[lo:hi], k)
    return [(int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]


def _fetch(b, want_stats, buffer=None):
    """-> (buffer, results, ops_off, words, stats): the packed cigars land in `buffer` when it is large enough, else in a new one
    (returned for the next chunk: 150 MB allocated and released per chunk are page faults and an munmap under the interpreter lock)."""
    res = b.results()
    need = int(res["n_ops"].sum())
    if buffer is None or buffer.size < need:
        _give(buffer)
        buffer = _take(4 * need).view(np.uint32)
    off, words, buffer = b.ops_packed_into(buffer)
    stats = b.align_stats() if want_stats else None
    return buffer, res, off, words, stats


def _rerun_overflowed(ctx, src, params, lo, res, off, words, stats, want_stats, tm):
    """Reads whose sparse posterior list overflowed its capacity (NPR_ERR_CAPACITY: a diffuse model can put up to
    1 / threshold pairs on a base) run again with a four times larger `max_pairs_per_base` until they fit, as
    Context.realign does; their results and cigars replace the failed ones."""
    from . import realign
    per_base = params.max_pairs_per_base if params.max_pairs_per_base > 0 else 6
    limit = int(1.0 / max(params.posterior_threshold, 1e-6)) + 1
    while per_base < limit:
        again = np.nonzero(res["status"] == realign.ERR_CAPACITY)[0]
        if not len(again):
            break
        per_base = min(4 * per_base, limit)
        p2 = realign.Params.from_buffer_copy(params)
        p2.max_pairs_per_base = per_base
        b = src.stage_records(ctx, p2, lo + again)
        try:
            tm["kernel_ms"] += b.run()
            b.finish()
            _, r2, o2, w2, s2 = _fetch(b, want_stats)
        finally:
            b.close()
        res[again] = r2
        if want_stats:
            stats[again] = s2
        pieces, prev = [], 0
        for k, i in enumerate(again):  # (a handful of reads: the loop is over them, not over the chunk)
            pieces += [words[off[prev]:off[i]], w2[o2[k]:o2[k + 1]]]
            prev = i + 1
        pieces.append(words[off[prev]:])
        words = np.concatenate(pieces)
        nops = off[1:] - off[:-1]
        nops[again] = o2[1:] - o2[:-1]
        off = np.concatenate([[0], np.cumsum(nops)]).astype(np.int64)
    return res, off, words, stats


def run_pipeline(src, params, lo, hi, ctxs, sink, want_stats=False, chunk_bases=None):
    """Records lo .. hi of `src` as a pipeline of chunks over the contexts `ctxs`, one thread per phase:

        stager (band planning + pack + H2D + device planner)  ->  DP pass  ->  finish (MEA chain + cigar on the device)
        ->  fetch + splice / format of the chunk's records  ->  `sink(block)` on the caller's thread, in record order
        (a sink that is done with a block when it returns says so by returning CONSUMED: the block's buffer is then used again;
        any other sink keeps what it was given).

    Chunk k uses context k mod len(ctxs) from its staging to its close, so len(ctxs) chunks are in flight and the stager
    runs that far ahead of the DP; the DP passes and MEA stages of different chunks take turns on the device's shared
    scratch (its mutex), everything else overlaps them.  A chunk the device cannot hold (NPR_ERR_NOMEM: the reference's
    per-read jobs have no such limit) is halved and staged again.  Returns (results[hi - lo], n_ops[hi - lo], stats or None,
    timings)."""
    from . import realign
    pending = list(reversed(chunk_bounds(src.lengths(), lo, hi, chunk_bases, len(ctxs))))  # a stack: splits go back on top
    # NPR_OPT_OVERLAP = 2: the MEA tables of a chunk off the device's shared scratch, so that the next chunk's DP pass starts when it
    # is staged and not when this chunk's MEA stage has given the scratch back (a kernel trace showed 10-15 ms per chunk of exactly
    # that wait: 414 -> 403 ms per 50 000 reads, 393 with GPU_MAX_HW_QUEUES=8).  Value 1 (the default since the end of round 5) also has
    # the DP launches leave half of every SIMD to the other chunks' staging and MEA kernels: with ONE slot per SIMD left free (rounds
    # 4-5) it bought nothing -- 80 registers and one wavefront hold none of the kernels that matter --, with four the job of 50 000
    # reads takes 352-361 ms instead of 367-372.  NPR_JOB_OVERLAP=0 / 1 / 2 picks one for an A/B run.
    # A rank held to a few host threads (NPR_HOST_THREADS, as bench.py sets it per rank) is bound by its host phases: the DP passes
    # keep the whole chip there (2 threads, 50 000 reads: 645 ms against 672).
    try:
        few_threads = 0 < int(os.environ.get("NPR_HOST_THREADS") or "0") < 4
    except ValueError:  # (the library reads the variable with atoi: anything else is "one thread" there)
        few_threads = True
    overlap = int(os.environ.get("NPR_JOB_OVERLAP") or ("2" if few_threads else "1")) if (len(ctxs) > 1 and len(pending) > 1) else 0  # (set but empty: the default)
    for c in ctxs:
        c.set_option(_lib.OPT_OVERLAP, overlap)
    n_planned = len(pending)
    tm = dict(stage_s=0.0, run_s=0.0, finish_s=0.0, fetch_s=0.0, format_s=0.0, kernel_ms=0.0, cells=0, trace=[])
    free = [threading.Semaphore(1) for _ in ctxs]
    q_run, q_fin, q_out, done = queue.Queue(), queue.Queue(), queue.Queue(), queue.Queue()
    stop = threading.Event()
    END = object()

    def note(label, t0, t1):
        tm[label + "_s"] += t1 - t0
        if TRACE:
            tm["trace"].append((label, t0, t1))

    def guarded(fn, upstream, downstream):
        def run():
            try:
                fn()
            except BaseException as e:  # handed to the caller's thread
                stop.set()
                done.put(e)
                # what the phases before this one still hand over is closed here: nobody else will look at this queue again
                while upstream is not None:
                    item = upstream.get()
                    if item is END:
                        break
                    item[3].close()
                    if item[5] is not None:
                        item[5].set()
                    if item[4]:
                        free[item[0]].release()
            finally:
                downstream.put(END)
        return run

    flushed = set()

    def stager():
        k = 0
        while pending and not stop.is_set():
            a, b_ = pending.pop()
            j = k % len(ctxs)
            while not free[j].acquire(timeout=0.2):
                if stop.is_set():
                    return
            t0 = time.perf_counter()
            try:
                batch = src.stage(ctxs[j], params, a, b_)
            except realign.NprError as e:
                free[j].release()
                if e.code != realign.ERR_NOMEM or b_ - a < 2:
                    raise
                if j not in flushed:  # once per context: the buffers it cached from earlier, differently sized batches may be what is in the
                    flushed.add(j)    # way (only this context's: nobody else touches it while the stager holds it)
                    ctxs[j].set_option(_lib.OPT_RELEASE_SCRATCH, 2)
                    pending.append((a, b_))
                    continue
                mid = (a + b_) // 2
                pending.extend([(mid, b_), (a, mid)])
                continue
            note("stage", t0, time.perf_counter())
            q_run.put((j, a, b_, batch, True, None))  # (last: the chunk's end gives the context back; gate: set when the batch is closed)
            k += 1

    def runner():
        while True:
            item = q_run.get()
            if item is END:
                return
            j, a, b_, batch, last, gate = item
            if stop.is_set():
                batch.close(), free[j].release()
                continue
            t0 = time.perf_counter()
            try:
                tm["kernel_ms"] += batch.run()
            except realign.NprError as e:
                batch.close()
                if e.code != realign.ERR_NOMEM or b_ - a < 2:
                    free[j].release()
                    raise
                # the device was full at launch (a kernel's private segment, the scratch regrown): the chunk's halves one after
                # the other in the same context, as the stager does for a chunk that does not fit at staging.  ONE AFTER THE OTHER all the
                # way: calls on one context must be serialised (include/nprealign.h), and the finisher and the fetcher work on the first
                # half's batch from their own threads -- the second half is staged when the fetcher has closed the first (its gate).
                try:
                    mid = (a + b_) // 2
                    for x, y, fin in ((a, mid, False), (mid, b_, True)):
                        half = src.stage(ctxs[j], params, x, y)
                        try:
                            tm["kernel_ms"] += half.run()
                        except BaseException:
                            half.close()
                            raise
                        gate = None if fin else threading.Event()
                        q_fin.put((j, x, y, half, fin, gate))
                        while gate is not None and not gate.wait(0.2):
                            if stop.is_set():  # a later phase failed (it closes what it was handed): nothing more is staged here
                                raise RuntimeError("pipeline stopped while a chunk's first half was in flight")
                        if gate is not None and stop.is_set():  # (a failing finisher SETS the gate on its way out: the wait above ends without having looked)
                            raise RuntimeError("pipeline stopped while a chunk's first half was in flight")
                except BaseException:
                    free[j].release()
                    raise
                note("run", t0, time.perf_counter())
                continue
            except BaseException:
                batch.close(), free[j].release()
                raise
            note("run", t0, time.perf_counter())
            q_fin.put(item)

    def finisher():
        while True:
            item = q_fin.get()
            if item is END:
                return
            j, a, b_, batch, last, gate = item
            if stop.is_set():
                batch.close()
                if gate is not None:
                    gate.set()
                if last:
                    free[j].release()
                continue
            t0 = time.perf_counter()
            try:
                batch.finish()
            except BaseException:
                batch.close()
                if gate is not None:
                    gate.set()
                if last:
                    free[j].release()
                raise
            note("finish", t0, time.perf_counter())
            q_out.put(item)

    def fetcher():
        words_buf = [None]
        while True:
            item = q_out.get()
            if item is END:
                _give(words_buf[0])
                return
            j, a, b_, batch, last, gate = item
            t0 = time.perf_counter()
            open_batch = [batch]
            try:  # the context goes back exactly once, whatever happens in between
                try:
                    if stop.is_set():
                        continue
                    words_buf[0], res, off, words, stats = _fetch(batch, want_stats, words_buf[0])
                    tm["cells"] += int(batch.stats()["cells"])
                    if (res["status"] == realign.ERR_CAPACITY).any():
                        batch.close(), open_batch.clear()  # (the reads that overflowed run again on this context)
                        res, off, words, stats = _rerun_overflowed(ctxs[j], src, params, a, res, off, words, stats, want_stats, tm)
                    t1 = time.perf_counter()
                    note("fetch", t0, t1)
                    # the records in two halves: the first is being written while the second is formatted (a job of one chunk -- a rank's
                    # share of a sharded set -- has nothing else to overlap its 8 ms of pwrite with)
                    m = (b_ - a) // 2
                    block = src.format_block(a, a + m, off[:m + 1], words[:int(off[m])])
                    done.put((block, None, None, None))
                    block = src.format_block(a + m, b_, off[m:] - off[m], words[int(off[m]):])
                    note("format", t1, time.perf_counter())
                    done.put((block, res, off[1:] - off[:-1], stats))
                    del block, res, off, words, stats
                finally:
                    # the batch goes after its block is on its way (releasing its host buffers takes 13 ms); the context after the batch
                    tc = time.perf_counter()
                    if open_batch:
                        batch.close()
                    if TRACE:
                        tm["trace"].append(("close", tc, time.perf_counter()))
            finally:
                if gate is not None:
                    gate.set()  # (the runner may stage the chunk's second half on this context now)
                if last:
                    free[j].release()

    threads = [threading.Thread(target=guarded(fn, up, down), daemon=True)
               for fn, up, down in ((stager, None, q_run), (runner, q_run, q_fin), (finisher, q_fin, q_out), (fetcher, q_out, done))]
    # The cyclic collector stays out of the job: a full collection walks every object of the process (millions once torch is
    # imported) with the interpreter lock held, tens of ms during which no phase can take its next chunk over.  (The 10-25 ms gaps
    # in round 4's traces turned out to be something else -- munmaps of the chunks' buffers under the same lock, see _take -- and
    # the job times the same with the collector on, NPR_JOB_GC=1; it is kept out because the pipeline makes no cycles worth collecting.)
    gc_held = os.environ.get("NPR_JOB_GC") is None and _gc_hold()
    for t in threads:
        t.start()
    parts, sink_s, error = [], 0.0, None
    finished = False
    try:
        while True:
            item = done.get()
            if item is END:
                finished = True
                break
            if isinstance(item, BaseException):
                error = error or item
                continue
            if error is None:
                block, res, nops, stats = item
                t0 = time.perf_counter()
                consumed = sink(block) is CONSUMED
                sink_s += time.perf_counter() - t0
                if res is not None:
                    parts.append((res, nops, stats))
                if TRACE:
                    tm["trace"].append(("sink", t0, time.perf_counter()))
                if consumed:
                    _give(block)  # (a pooled buffer goes back; anything else is released now, not when the next block is waiting to be written)
                del block, item
    finally:
        if not finished:  # the sink failed on this thread (a full disk): the phases stop, what is in flight is closed, nothing keeps a context
            stop.set()
            while done.get() is not END:
                pass
        for t in threads:
            t.join()
        if gc_held:
            _gc_release()
    if error is not None:
        raise error
    n = hi - lo
    results = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0, dtype=_lib.RESULT_DTYPE)
    n_ops = np.concatenate([p[1] for p in parts]) if parts else np.zeros(0, dtype=np.int64)
    stats = np.concatenate([p[2] for p in parts]) if (parts and want_stats) else (np.zeros((0, _lib.STATS_WORDS), dtype=np.int32) if want_stats else None)
    assert len(results) == n
    if not TRACE:
        del tm["trace"]
    tm["sink_s"] = sink_s
    tm["chunks"] = len(parts)
    tm["planned_chunks"] = n_planned
    tm["in_flight"] = len(ctxs)
    return results, n_ops, stats, tm


# ---------------------------------------------------------------------------------------------------------
# the job: shard, pipeline, write, gather
# ---------------------------------------------------------------------------------------------------------

CONSUMED = "consumed"  # what a sink of run_pipeline returns when it is done with the block it was given (its buffer is used again)
SOLO = "solo"  # as `group`: this process alone runs the whole job although torch.distributed is initialised (bench.py's one-rank
               # reference run of the strong-scaling job inside an N-rank launch)


def _dist_state(group):
    import torch.distributed as dist
    multi = group is not SOLO and dist.is_available() and dist.is_initialized()
    if not multi:
        return None, 1, 0
    return dist, dist.get_world_size(group), dist.get_rank(group)


def _collective_device(dist, group, gpu):
    """Where the tensors of the job's two small collectives live: on the rank's GPU under nccl (= RCCL), on the host else."""
    import torch
    if dist is not None and str(dist.get_backend(group)) == "nccl":
        return torch.device("cuda", gpu)
    return torch.device("cpu")


def default_gpu():
    """The GPU of this rank: LOCAL_RANK when torch.distributed runs one process per GPU, else device 0."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() > 1:
        try:
            return int(os.environ.get("LOCAL_RANK", torch.cuda.current_device()))
        except ValueError:
            return torch.cuda.current_device()
    return 0


def write_summary_xml(path, status, score, nops, cells=None):
    """Summary of the job for rank 0's report (the reference's per-experiment XMLs are built from exactly these per-read
    scalars, e.g. alignmentUncertainty.py:59-64)."""
    ok = np.asarray(status) == 0
    root = ET.Element("realignSummary")
    root.set("reads", str(len(status)))
    root.set("failedReads", str(int((~ok).sum())))
    root.set("averagePosteriorMatchProbabilityPerRead", repr(float(np.mean(np.asarray(score)[ok])) if ok.any() else float("nan")))
    root.set("cigarOps", str(int(np.sum(nops))))
    if cells is not None:
        root.set("cells", str(int(cells)))
    ET.ElementTree(root).write(path)


def run_source(src, params, bounds, out_path, ctxs=None, gpu=None, group=None, want_stats=False, chunk_bases=None, workers=None,
               coll_device=None):
    """The job on this rank (collective: every rank of the process group calls it; without torch.distributed initialised it
    is the one-GPU job).  `src` holds this rank's view of the records, `bounds[r] .. bounds[r + 1]` the range of rank r in
    `src`'s numbering.  Output: `out_path` = src.header + every rank's block in rank order.

    One rank: blocks are written as they complete, under the DP of the chunks behind them.  Several ranks: a rank's offset
    is known once the ranks before it know their sizes, so rank 0 writes as it goes and the others keep their blocks (a
    rank's share of the text: tens of MB) and write them after the all_gather of the sizes.  Only the per-read results are
    gathered to rank 0 (RCCL over xGMI under backend nccl).  Returns a dict: on rank 0 `results` (structured array in input order),
    `n_ops`, `stats` (if asked for), `timings`; on the others `timings` only."""
    import torch
    _lib.want_hw_queues()  # (a deployment setting, INTEGRATION.md: asked for here, not when the binding is imported)
    dist, world, rank = _dist_state(group)
    gpu = default_gpu() if gpu is None else gpu
    dev = torch.device(coll_device) if coll_device is not None else _collective_device(dist, group, gpu)
    ctxs = ctxs or contexts(gpu, workers or WORKERS)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    header = src.header
    t_begin = time.perf_counter()
    fd = None
    kept = []
    state = dict(off=len(header))
    if out_path is not None and rank == 0:
        fd = os.open(out_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        os.pwrite(fd, header, 0)

    def sink(block):
        if out_path is None:
            return CONSUMED
        if rank == 0:
            os.pwrite(fd, memoryview(block), state["off"])
            state["off"] += len(block)
            return CONSUMED
        kept.append(block)  # (written after the all_gather of the sizes: the block stays this rank's)

    try:
        results, n_ops, stats, tm = run_pipeline(src, params, lo, hi, ctxs, sink, want_stats=want_stats, chunk_bases=chunk_bases)
        t0 = time.perf_counter()
        if out_path is not None and dist is not None:
            mine = state["off"] - len(header) if rank == 0 else sum(len(b) for b in kept)
            t = torch.tensor([mine], dtype=torch.int64, device=dev)
            got = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(got, t, group=group)
            sizes = [int(g.item()) for g in got]
            if rank != 0:
                fd = os.open(out_path, os.O_WRONLY)  # rank 0 created it before its first chunk; this is after everyone's last
                at = len(header) + sum(sizes[:rank])
                for b in kept:
                    os.pwrite(fd, memoryview(b), at)
                    at += len(b)
        tm["write_tail_s"] = time.perf_counter() - t0
    finally:
        if fd is not None:
            os.close(fd)
    out = dict(timings=tm)
    # the one gather: per-read results for the summary
    t0 = time.perf_counter()
    if dist is not None:
        payload = [results.view(np.uint8).reshape(-1), n_ops.astype(np.int64).view(np.uint8)]
        if want_stats:
            payload.append(np.ascontiguousarray(stats).view(np.uint8).reshape(-1))
        head = np.array([len(results)], dtype=np.int64).view(np.uint8)
        got = npd.gather_to_root(np.concatenate([head] + payload), device=dev, group=group)
        if rank == 0:
            rs, ns, ss = [], [], []
            for g in got:
                k = int(g[:8].view(np.int64)[0])
                a = 8
                rs.append(g[a:a + k * _lib.RESULT_DTYPE.itemsize].view(_lib.RESULT_DTYPE))
                a += k * _lib.RESULT_DTYPE.itemsize
                ns.append(g[a:a + 8 * k].view(np.int64))
                a += 8 * k
                if want_stats:
                    ss.append(g[a:a + 4 * _lib.STATS_WORDS * k].view(np.int32).reshape(k, _lib.STATS_WORDS))
            results, n_ops = np.concatenate(rs), np.concatenate(ns)
            stats = np.concatenate(ss) if want_stats else None
        dist.barrier(group=group)  # every rank's block is on disk when rank 0 returns
    tm["gather_s"] = time.perf_counter() - t0
    tm["wall_s"] = time.perf_counter() - t_begin
    _trim_host_pool(_HOST_POOL_BYTES // 4)  # a long-lived host application does not keep a job's gigabyte of chunk buffers
    if TRACE:
        tm["trace"] += [("source", t_begin, t_begin), ("gather", t0, time.perf_counter())]
    if rank == 0:
        out.update(results=results, n_ops=n_ops, stats=stats, sam=out_path)
    return out


def realign_sam_file(samFile, outputSamFile, referenceFastaFile, hmm=None, gapGamma=0.5, matchGamma=0.0, params=None, group=None,
                     gpu=None, want_stats=False, model_slot=0, chunk_bases=None, workers=None, coll_device=None, set_models=True):
    """Files -> file: every record of `samFile` that has a reference realigned against `referenceFastaFile`, written to
    `outputSamFile` with only its CIGAR replaced, same order, header copied (realignSamFile2TargetFn + realignCigarTargetFn +
    realignSamFile3TargetFn, utils.py:557-609).  Under an initialised torch.distributed process group the records shard over
    the ranks (collective call).  `hmm`: a nanopore_amd.hmm.Hmm, a model file path, or None for the stock model;
    `params`: npr_params overriding the reference's call parameters (anchors +- 10, trim 14, split 3000) -- the bench's
    fixed-band configs.  Returns run_source's dict, plus `records` = the number of records kept (rank 0)."""
    from . import ingest, realign
    from .hmm import Hmm
    dist, world, rank = _dist_state(group)
    gpu = default_gpu() if gpu is None else gpu
    t0 = time.perf_counter()
    sam = ingest.SamText(samFile)
    fasta = ingest.FastaTable(referenceFastaFile)
    t1 = time.perf_counter()
    bounds = npd.shard_ranges(sam.line_lengths(), world)  # a record's length in the file ~ its read's length
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    fields = sam.parse(lo, hi)
    span = sam.span[lo:hi]
    # every line is checked BEFORE anything is dropped (a truncated SAM or one without @SQ lines must not yield a successful job
    # with fewer records); samIterator drops the records whose RNAME is "*", and only those (utils.py:287-293)
    keep = sam.records_with_a_reference(fields, span)
    if not keep.all():
        fields, span = fields[keep], span[keep]
    src = SamSource(sam, fasta, span, fields, model_slot=None if not model_slot else np.full(len(fields), model_slot, dtype=np.int32))
    t2 = time.perf_counter()
    ctxs = contexts(gpu, workers or WORKERS)
    if set_models:
        model = Hmm.loadHmm(hmm) if isinstance(hmm, str) else hmm
        for c in ctxs:
            c.set_hmm(model, slot=model_slot)
    if params is None:
        params = realign.make_params(band_mode=realign.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=3000,
                                     gap_gamma=gapGamma, match_gamma=matchGamma, mode=realign.MODE_REALIGN)
    local = np.array([0] * (rank + 1) + [len(fields)] * (world - rank), dtype=np.int64)  # src holds this rank's records only
    out = run_source(src, params, local, outputSamFile, ctxs=ctxs, gpu=gpu, group=group, want_stats=want_stats, chunk_bases=chunk_bases,
                     coll_device=coll_device)
    out["timings"].update(index_s=t1 - t0, parse_s=t2 - t1)
    out["timings"]["wall_s"] += t2 - t0
    if TRACE:
        out["timings"]["trace"] += [("index", t0, t1), ("parse", t1, t2), ("return", time.perf_counter(), time.perf_counter())]
    if rank == 0:
        out["records"] = len(out["results"])
    return out


def run_job(ctx, params, w, out_dir=None, work=None, model_slot=None, device=None, group=None, chunk_bases=None, workers=None):
    """The job over a resident synthetic workload dict (nanopore_amd.synth): `bench.py --workload c3` and the two-rank test.
    `ctx` is the first of the rank's contexts (more are made on its GPU as the pipeline needs them).  Returns on rank 0 a dict
    with status / score / n_ops in input order, the output paths and the rank's timings; on other ranks the timings only."""
    dist, world, rank = _dist_state(group)
    src = SynthSource(w, model_slot=model_slot)
    if work is None:
        work = src.lengths()
    bounds = npd.shard_ranges(work, world)
    workers = workers or WORKERS
    ctxs = [ctx] + [c for c in contexts(ctx.device, workers) if c is not ctx][:workers - 1]
    for c in ctxs[1:]:  # the extra contexts run the same models
        c.copy_models_from(ctx)
    sam_path = None
    if out_dir is not None:
        if rank == 0:
            os.makedirs(out_dir, exist_ok=True)
        if dist is not None:
            dist.barrier(group=group)
        sam_path = os.path.join(out_dir, "realigned.sam")
    out = run_source(src, params, bounds, sam_path, ctxs=ctxs, gpu=ctx.device, group=group, chunk_bases=chunk_bases, coll_device=device)
    if rank != 0:
        return out
    res = out["results"]
    out["status"], out["score"] = res["status"].astype(np.int64), res["score"].astype(np.float64)
    if out_dir is not None:
        out["xml"] = os.path.join(out_dir, "summary.xml")
        write_summary_xml(out["xml"], out["status"], out["score"], out["n_ops"], cells=int(res["cells"].sum()))
    return out
