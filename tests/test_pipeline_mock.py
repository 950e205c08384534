"""The per-rank pipeline of the files -> file job (nanopore_amd/job.py::run_pipeline: stage | DP | finish | fetch + format, one thread
per phase, chunks in flight on a pool of contexts) over stand-in batches, no GPU: record order, the halving of a chunk the device
cannot hold (at staging and at launch), and the error paths -- whatever fails, every staged batch is closed, every context is given
back exactly once and the phase threads are gone.  Reference: one jobTree job per record (nanopore/analyses/utils.py:565-570) has
no such state to clean up; a failed job fails the tree (pipeline.py:209-210), here the exception reaches the caller."""
import threading

import numpy as np
import pytest

from nanopore_amd import _lib, job, realign


class FakeCtx(object):
    def __init__(self):
        self.open = 0
        self.peak = 0

    def set_option(self, option, value):
        self.options = getattr(self, "options", {})
        self.options[option] = value


class FakeBatch(object):
    def __init__(self, src, ctx, a, b):
        self.src, self.ctx, self.a, self.b = src, ctx, a, b
        ctx.open += 1
        ctx.peak = max(ctx.peak, ctx.open)
        src.staged.append((a, b))
        self.closed = False

    def run(self):
        if self.src.fail_run == "nomem" and self.b - self.a > self.src.max_run:
            raise realign.NprError(realign.ERR_NOMEM, "npr_batch_run", "device full")
        if self.src.fail_run == "boom":
            raise RuntimeError("launch failed")
        return 1.0

    def finish(self):
        if self.src.fail_finish:
            raise RuntimeError("finish failed")

    def results(self):
        r = np.zeros(self.b - self.a, dtype=_lib.RESULT_DTYPE)
        r["score"] = np.arange(self.a, self.b)
        return r

    def ops_packed(self):
        n = self.b - self.a
        return np.arange(n + 1, dtype=np.int64), np.arange(self.a, self.b).astype(np.uint32)

    def ops_packed_into(self, buffer):
        off, words = self.ops_packed()
        if buffer is None or buffer.size < len(words):
            buffer = np.empty(len(words) + 8, dtype=np.uint32)
        buffer[:len(words)] = words
        return off, buffer[:len(words)], buffer

    def stats(self):
        return {"cells": 10 * (self.b - self.a)}

    def close(self):
        assert not self.closed, "a batch is closed twice"
        self.closed = True
        self.ctx.open -= 1
        self.src.closed += 1


class FakeSrc(object):
    def __init__(self, n, max_stage=1 << 30, max_run=1 << 30, fail_run=None, fail_finish=False):
        self.n, self.max_stage, self.max_run, self.fail_run, self.fail_finish = n, max_stage, max_run, fail_run, fail_finish
        self.staged, self.closed = [], 0

    def lengths(self):
        return np.full(self.n, 1000, dtype=np.int64)

    def stage(self, ctx, params, lo, hi):
        if hi - lo > self.max_stage:
            raise realign.NprError(realign.ERR_NOMEM, "npr_batch_create", "device full")
        return FakeBatch(self, ctx, lo, hi)

    def format_block(self, lo, hi, ops_off, words):
        assert list(words) == list(range(lo, hi))
        return np.arange(lo, hi, dtype=np.int64).tobytes()


def _run(src, ctxs, sink=None, chunk_bases=100000):
    blocks = []
    params = realign.make_params()
    out = job.run_pipeline(src, params, 0, src.n, ctxs, sink or blocks.append, chunk_bases=chunk_bases)
    return out, blocks


def _all_back(src, ctxs, threads_before):
    assert src.closed == len(src.staged) and all(c.open == 0 for c in ctxs)
    assert threading.active_count() <= threads_before


@pytest.mark.parametrize("kw", [dict(), dict(max_stage=60), dict(fail_run="nomem", max_run=60)], ids=["plain", "nomem_at_staging", "nomem_at_launch"])
def test_blocks_come_in_record_order_and_chunks_that_do_not_fit_are_halved(kw):
    before = threading.active_count()
    ctxs = [FakeCtx() for _ in range(3)]
    src = FakeSrc(1000, **kw)
    (res, nops, stats, tm), blocks = _run(src, ctxs)
    got = np.frombuffer(b"".join(blocks), dtype=np.int64)
    assert list(got) == list(range(1000)) and list(res["score"]) == list(range(1000)) and tm["cells"] == 10000
    # one batch per context at any time -- also while the halves of a chunk the launch refused pass through: the second half is staged
    # when the fetcher has closed the first (calls on one npr_ctx must be serialised, include/nprealign.h; round 4's advisor found the
    # runner staging it while the finisher and the fetcher were still working on the first)
    assert all(c.peak == 1 for c in ctxs)
    if kw:
        assert max(b - a for a, b in src.staged if not kw.get("fail_run") or True) >= 50 and len(src.staged) > 10
    _all_back(src, ctxs, before)


@pytest.mark.parametrize("threads,want", [(None, 1), ("16", 1), ("2", 2)])
def test_contexts_of_a_pipelined_job_leave_room_beside_their_dp_passes(monkeypatch, threads, want):
    """NPR_OPT_OVERLAP of a job of several chunks: 1 (own MEA tables, DP launches that leave half of every SIMD to the other chunks' kernels),
    2 (the tables only) for a rank held to a few host threads; 0 for a job of one chunk (nothing to run beside)."""
    if threads is None:
        monkeypatch.delenv("NPR_HOST_THREADS", raising=False)
    else:
        monkeypatch.setenv("NPR_HOST_THREADS", threads)
    monkeypatch.delenv("NPR_JOB_OVERLAP", raising=False)
    ctxs = [FakeCtx() for _ in range(3)]
    _run(FakeSrc(1000), ctxs)
    assert all(c.options[_lib.OPT_OVERLAP] == want for c in ctxs)
    one = [FakeCtx() for _ in range(3)]
    _run(FakeSrc(40), one)
    assert all(c.options[_lib.OPT_OVERLAP] == 0 for c in one)


@pytest.mark.parametrize("kw,exc", [(dict(fail_run="boom"), RuntimeError), (dict(fail_finish=True), RuntimeError),
                                    (dict(max_stage=0), realign.NprError)], ids=["launch", "finish", "staging"])
def test_a_failing_phase_reaches_the_caller_and_leaves_nothing_behind(kw, exc):
    before = threading.active_count()
    ctxs = [FakeCtx() for _ in range(3)]
    src = FakeSrc(1000, **kw)
    with pytest.raises(exc):
        _run(src, ctxs)
    _all_back(src, ctxs, before)


def test_a_failing_sink_stops_the_phases():
    """os.pwrite failing on the caller's thread (a full disk): the exception is the caller's, the phase threads stop, every batch in
    flight is closed and every context given back."""
    before = threading.active_count()
    ctxs = [FakeCtx() for _ in range(3)]
    src = FakeSrc(2000)
    seen = []

    def sink(block):
        seen.append(len(block))
        if len(seen) == 2:
            raise OSError(28, "No space left on device")

    with pytest.raises(OSError):
        _run(src, ctxs, sink=sink)
    _all_back(src, ctxs, before)
    # the contexts can be used again at once: nothing holds their semaphores
    src2 = FakeSrc(300)
    (res, _, _, _), blocks = _run(src2, ctxs)
    assert len(res) == 300


def test_host_buffers_go_round():
    """job._take / job._give: the per-chunk host buffers of a job (guide operations, packed cigars, formatted records) are kept from
    chunk to chunk; a view finds its way back to its buffer, foreign arrays are not adopted, the pool is bounded and close_contexts()
    empties it."""
    job.close_contexts()
    a = job._take(1000)
    assert a.dtype == np.uint8 and a.nbytes >= 1000 and a.nbytes % 4096 == 0
    job._give(a.view(np.uint32)[3:50])
    assert job._take(500) is a and len(job._host_pool) == 0
    job._give(a), job._give(a)                                    # (given back twice: kept once)
    assert len(job._host_pool) == 1
    big = job._take(10 * a.nbytes)
    assert big is not a and big.nbytes >= 10 * a.nbytes           # the pooled one is too small: a new one
    job._give(np.zeros(4096 * 3 + 1, dtype=np.uint8)), job._give(b"bytes"), job._give(None), job._give(np.zeros(4096, dtype=np.int32))
    assert len(job._host_pool) == 1
    job._give(big)
    assert job._take(a.nbytes + 1) is big and job._take(1) is a   # the smallest that fits
    for _ in range(2 * job._HOST_POOL_MAX):
        job._give(np.empty(4096, dtype=np.uint8))
    assert len(job._host_pool) == job._HOST_POOL_MAX
    job.close_contexts()
    assert len(job._host_pool) == 0


def test_collector_is_held_off_once_for_jobs_side_by_side():
    import gc
    assert gc.isenabled()
    job._gc_hold(), job._gc_hold()
    assert not gc.isenabled()
    job._gc_release()
    assert not gc.isenabled()          # the other job still runs
    job._gc_release()
    assert gc.isenabled()
    gc.disable()
    try:
        job._gc_hold(), job._gc_release()
        assert not gc.isenabled()      # it was off when the job came: it stays off
    finally:
        gc.enable()


def test_chunks_are_capped_by_bases_as_well_as_floored_by_reads():
    short = np.full(50000, 8000, dtype=np.int64)     # BASELINE configs[2]: eight chunks of 6 250 reads
    assert len(job.chunk_bounds(short, 0, len(short))) == 8
    assert len(job.chunk_bounds(short, 0, 12500)) == 3 and len(job.chunk_bounds(short, 0, 6250)) == 1  # a rank's share at N = 4 / 8
    long_ = np.full(30000, 80000, dtype=np.int64)    # 30 000 reads of 80 kb: 12 288 reads would be ten chunks' worth of bases each
    cuts = job.chunk_bounds(long_, 0, len(long_))
    assert max(b - a for a, b in cuts) * 80000 <= 4.2 * job.CHUNK_BASES and cuts[0][0] == 0 and cuts[-1][1] == len(long_)
    assert all(a2 == b1 for (_, b1), (a2, _) in zip(cuts[:-1], cuts[1:]))


def test_host_pool_is_bounded_by_bytes_and_trimmed(monkeypatch):
    job.close_contexts()
    monkeypatch.setattr(job, "_HOST_POOL_BYTES", 10 * 4096)
    for _ in range(4):
        job._give(np.empty(4 * 4096, dtype=np.uint8))
    assert sum(b.nbytes for b in job._host_pool) <= 10 * 4096 and len(job._host_pool) == 2
    job._trim_host_pool(5 * 4096)
    assert len(job._host_pool) == 1
    job.close_contexts()
