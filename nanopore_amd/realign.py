"""Batched realignment API over the C ABI (libnprealign.so).

`Context.realign()` is the in-process replacement of the reference's per-read fan-out
(nanopore/analyses/utils.py:557-609: one jobTree job + one `cactus_realign` process per SAM record):
all reads of a SAM file go to the GPU in one call.  Device work only -- no CPU fallback.
"""
import contextlib
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import (BAND_ANCHOR, BAND_FIXED, ERR_CAPACITY, ERR_NOMEM, MODE_ALL_POSTERIORS, MODE_EXPECTATIONS, MODE_REALIGN, MODE_RESCORE_ORIGINAL, NprError,
                   Params, ptr)


def make_params(band_mode=BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=3000,
                fixed_width=0, gap_gamma=0.5, match_gamma=0.0, posterior_threshold=0.01, mode=MODE_REALIGN,
                max_pairs_per_base=0):
    """Defaults are the realign call string of nanopore/analyses/utils.py:587 and
    AbstractMapper.realignSamFile's gammas (nanopore/mappers/abstractMapper.py:25)."""
    return Params(band_mode, diagonal_expansion, constraint_trim, split_threshold, fixed_width, gap_gamma,
                  match_gamma, posterior_threshold, mode, max_pairs_per_base)


def _csr(seqs):
    """list of bytes/str -> (uint8 buffer, int64 offsets)."""
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    buf = np.empty(int(off[-1]), dtype=np.uint8)
    for i, s in enumerate(seqs):
        if isinstance(s, str):
            s = s.encode("ascii")
        buf[off[i]:off[i + 1]] = np.frombuffer(s, dtype=np.uint8) if not isinstance(s, np.ndarray) else s
    return buf, off


def _csr_ops(guides):
    lens = np.fromiter((len(g) for g in guides), dtype=np.int64, count=len(guides))
    off = np.zeros(len(guides) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    ops = np.zeros((int(off[-1]), 2), dtype=np.int32)
    for i, g in enumerate(guides):
        if len(g):
            ops[off[i]:off[i + 1]] = np.asarray(g, dtype=np.int32).reshape(-1, 2)
    return ops, off


class Batch(object):
    """A staged batch: inputs resident in HBM after construction."""

    def __init__(self, ctx, params, ref, ref_off, read, read_off, guide_ops, guide_off, model_slot=None,
                 ref_index=None, guide_start=None, read_end=None):
        self._L = _lib.load()
        self.ctx = ctx
        n_refs = len(ref_off) - 1
        self._keep = (ref, ref_off, read, read_off, guide_ops, guide_off, model_slot, ref_index, guide_start, read_end)
        h = C.c_void_p()
        if read_end is not None:  # reads scattered in `read` (the SEQ fields of a mapped SAM text): read_off = their starts
            self.n_reads = len(read_off)
            rc = self._L.npr_batch_create_spans(ctx._h, C.byref(params), self.n_reads, n_refs, ptr(ref), ptr(ref_off),
                                                ptr(ref_index), ptr(read), ptr(read_off), ptr(read_end), ptr(guide_ops), ptr(guide_off),
                                                ptr(guide_start), ptr(model_slot), C.byref(h))
        else:
            self.n_reads = len(read_off) - 1
            rc = self._L.npr_batch_create_at(ctx._h, C.byref(params), self.n_reads, n_refs, ptr(ref), ptr(ref_off),
                                             ptr(ref_index), ptr(read), ptr(read_off), ptr(guide_ops), ptr(guide_off),
                                             ptr(guide_start), ptr(model_slot), C.byref(h))
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_create", ctx.last_error())
        self._h = h
        ctx._open.add(self)

    def class_stats(self):
        """(tasks, cells) per kernel class (include/nprealign.h: npr_batch_class_stats)."""
        t = np.zeros(32, dtype=np.int64)
        c = np.zeros(32, dtype=np.int64)
        k = self._L.npr_batch_class_stats(self._h, ptr(t), ptr(c), 32)
        return t[:k], c[:k]

    def segment_arith(self):
        """-> (seg_off[n+1], arith): per segment of every read, in read order, the device arithmetic it ran in (0: one
        exponent per cell, 1: one per anti-diagonal row; include/nprealign.h: npr_batch_segment_arith)."""
        off = np.zeros(self.n_reads + 1, dtype=np.int64)
        rc = self._L.npr_batch_segment_arith(self._h, ptr(off), None, 0)
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_segment_arith")
        ar = np.zeros(max(int(off[-1]), 1), dtype=np.int32)
        self._L.npr_batch_segment_arith(self._h, ptr(off), ptr(ar), len(ar))
        return off, ar[:int(off[-1])]

    def stats(self):
        st = _lib.BatchStats()
        self._L.npr_batch_get_stats(self._h, C.byref(st))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def run(self):
        """Device DP pass (forward + backward + posterior extraction).  Returns kernel milliseconds
        measured with HIP events on the library's stream."""
        ms = C.c_float(0)
        rc = self._L.npr_batch_run(self._h, C.byref(ms))
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_run", self.ctx.last_error())
        return ms.value

    def finish(self):
        rc = self._L.npr_batch_finish(self._h)
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_finish", self.ctx.last_error())

    def results(self):
        out = np.zeros(self.n_reads, dtype=_lib.RESULT_DTYPE)
        assert out.dtype.itemsize == C.sizeof(_lib.ReadResult)
        rc = self._L.npr_batch_results(self._h, ptr(out))
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_results")
        return out

    def ops(self):
        """-> (ops_off[n+1], ops[k,2]) CSR of output cigars."""
        off = np.zeros(self.n_reads + 1, dtype=np.int64)
        rc = self._L.npr_batch_ops(self._h, ptr(off), None, 0)
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_ops")
        ops = np.zeros((int(off[-1]), 2), dtype=np.int32)
        rc = self._L.npr_batch_ops(self._h, ptr(off), ptr(ops), len(ops))
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_ops")
        return off, ops

    def ops_packed(self):
        """Output cigars, one uint32 per op (length << 2 | op): (offsets[n+1], words)."""
        off, words, _ = self.ops_packed_into(None, slack=False)
        return off, words

    def ops_packed_into(self, buffer, slack=True):
        """ops_packed for a caller that fetches batch after batch: -> (offsets, words, buffer) -- the words are a view of `buffer`
        (a uint32 array or None) when it is large enough, else of a new, larger one, which is returned for the next call."""
        off = np.zeros(self.n_reads + 1, dtype=np.int64)
        rc = self._L.npr_batch_ops_packed(self._h, ptr(off), None, 0)
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_ops_packed", self.ctx.last_error())
        total = int(off[-1])
        if buffer is None or buffer.size < max(total, 1):
            buffer = np.empty(max(total + (total // 4 if slack else 0), 1), dtype=np.uint32)  # (filled by the call: no memset first)
        rc = self._L.npr_batch_ops_packed(self._h, ptr(off), ptr(buffer), total)
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_ops_packed", self.ctx.last_error())
        return off, buffer[:total], buffer

    def debug_set_pairs(self, read, x, y, p, task_status=0):
        """TEST HOOK (include/nprealign.h npr_batch_debug_set_pairs): replaces on the device the posterior pairs the DP pass left for `read`."""
        x = np.ascontiguousarray(x, np.int32)
        y = np.ascontiguousarray(y, np.int32)
        p = np.ascontiguousarray(p, np.float32)
        rc = self._L.npr_batch_debug_set_pairs(self._h, int(read), ptr(x), ptr(y), ptr(p), len(x), int(task_status))
        if rc != 0:
            raise NprError(rc, "npr_batch_debug_set_pairs")

    def pairs(self):
        """-> (pair_off[n+1], x, y, p) sparse posterior match probabilities sorted by (x, y)."""
        off = np.zeros(self.n_reads + 1, dtype=np.int64)
        rc = self._L.npr_batch_pairs(self._h, ptr(off), None, None, None, 0)
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_pairs")
        k = int(off[-1])
        x = np.zeros(k, dtype=np.int32)
        y = np.zeros(k, dtype=np.int32)
        p = np.zeros(k, dtype=np.float32)
        rc = self._L.npr_batch_pairs(self._h, ptr(off), ptr(x), ptr(y), ptr(p), k)
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_pairs")
        return off, x, y, p

    def base_expectations(self, ref_lengths, use=None):
        """Posterior-weighted base counts per reference position of the (selected) reads, scattered on the device
        (include/nprealign.h: npr_batch_base_expectations): (expect[sum(ref_lengths), 4], seen[sum(ref_lengths)])."""
        ref_lengths = np.ascontiguousarray(ref_lengths, dtype=np.int64)
        rows = int(ref_lengths.sum())
        expect = np.zeros((rows, 4), dtype=np.float64)
        seen = np.zeros(rows, dtype=np.uint8)
        u = None if use is None else np.ascontiguousarray(use, dtype=np.uint8)
        rc = self._L.npr_batch_base_expectations(self._h, ptr(u), len(ref_lengths), ptr(ref_lengths), ptr(expect), ptr(seen))
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_base_expectations", self.ctx.last_error())
        return expect, seen.astype(bool)

    def plan_check(self):
        """Tasks whose device-made band rows / schedules / stripe tables differ from the host planner's (test aid;
        include/nprealign.h: npr_batch_plan_check)."""
        rc = self._L.npr_batch_plan_check(self._h, ptr(self._keep[4]))
        if rc < 0:
            raise NprError(int(rc), "npr_batch_plan_check", self.ctx.last_error())
        return int(rc)

    def align_stats(self):
        """Per-read reductions over the aligned pairs of the cigars finish() produced, computed where they lie
        (include/nprealign.h: npr_batch_align_stats): int32 array [n_reads, STATS_WORDS]."""
        out = np.zeros((self.n_reads, _lib.STATS_WORDS), dtype=np.int32)
        rc = self._L.npr_batch_align_stats(self._h, ptr(out))
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_align_stats", self.ctx.last_error())
        return out

    def expectations(self):
        """Baum-Welch E-step with the installed models: (T_exp[slots,25], E_exp[slots,80], loglik[slots], kernel ms)."""
        T = np.zeros((_lib.MAX_MODELS, 25))
        E = np.zeros((_lib.MAX_MODELS, 80))
        ll = np.zeros(_lib.MAX_MODELS)
        ms = C.c_float(0)
        rc = self._L.npr_batch_expectations(self._h, ptr(T), ptr(E), ptr(ll), C.byref(ms))
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_expectations", self.ctx.last_error())
        return T, E, ll, ms.value

    def dense(self, read_index, cells):
        fv = np.zeros(cells, dtype=np.float32)
        fe = np.zeros(cells, dtype=np.int32)
        bv = np.zeros(cells, dtype=np.float32)
        be = np.zeros(cells, dtype=np.int32)
        rc = self._L.npr_batch_dense(self._h, read_index, ptr(fv), ptr(fe), ptr(bv), ptr(be), cells)
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_dense", self.ctx.last_error())
        return fv, fe, bv, be

    def rs_forward(self, read_index, cells):
        """include/nprealign.h: npr_batch_rs_forward -- the forward match rows k_dp_rs stores, (value, row exponent) in band order."""
        fv = np.zeros(cells, dtype=np.float32)
        fe = np.zeros(cells, dtype=np.int32)
        rc = self._L.npr_batch_rs_forward(self._h, read_index, ptr(fv), ptr(fe), cells)
        if rc != _lib.OK:
            raise NprError(rc, "npr_batch_rs_forward", self.ctx.last_error())
        return fv, fe

    def close(self):
        if getattr(self, "_h", None):
            self._L.npr_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context(object):
    """One realigner context = one GPU + one HIP stream.  Raises if no gfx950 device is usable."""

    def __init__(self, device=0):
        self._L = _lib.load()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self._L.npr_create(device, C.byref(h), err, 512)
        if rc != _lib.OK:
            raise NprError(rc, "npr_create", err.value.decode(errors="replace"))
        self._h = h
        self.device = device
        self._models = {}  # slot -> (T, E) or None, as installed: what another context on the same GPU copies
        self._open = weakref.WeakSet()  # the batches staged on this context and not closed yet: close() closes them first

    def set_option(self, option, value):
        """include/nprealign.h: npr_ctx_option (e.g. _lib.OPT_OVERLAP for a context of a pipelined job)."""
        rc = self._L.npr_ctx_option(self._h, option, int(value))
        if rc != _lib.OK:
            raise NprError(rc, "npr_ctx_option", self.last_error())

    def release_scratch(self):
        """NPR_OPT_RELEASE_SCRATCH: the device's forward scratch and this context's cached buffers go back to the driver."""
        self.set_option(_lib.OPT_RELEASE_SCRATCH, 1)

    @contextlib.contextmanager
    def options(self, **kw):
        """The test / bring-up switches of include/nprealign.h by name (_lib.OPTIONS), set for the body and back to 0 after it:
        `with ctx.options(arith=_lib.ARITH_CELL): ...`.  None changes a result."""
        for k, v in kw.items():
            self.set_option(_lib.OPTIONS[k], v)
        try:
            yield self
        finally:
            for k in kw:
                self.set_option(_lib.OPTIONS[k], 0)

    def copy_models_from(self, other):
        """Installs the models `other` holds (a second context of a pipelined job runs the same ones)."""
        for slot, m in other._models.items():
            self._set(slot, m)

    def _set(self, slot, m):
        rc = self._L.npr_set_hmm(self._h, slot, None, None) if m is None else self._L.npr_set_hmm(self._h, slot, ptr(m[0]), ptr(m[1]))
        if rc != _lib.OK:
            raise NprError(rc, "npr_set_hmm", self.last_error())
        self._models[slot] = m

    def last_error(self):
        return self._L.npr_last_error(self._h).decode(errors="replace")

    def set_hmm(self, hmm=None, slot=0):
        """hmm: object with .transitions (25) and .emissions (80) (nanopore_amd.hmm.Hmm), or None for
        the stock model used when the reference passes no --loadHmm."""
        if hmm is None:
            return self._set(slot, None)
        T = np.ascontiguousarray(hmm.transitions, dtype=np.float64)
        E = np.ascontiguousarray(hmm.emissions, dtype=np.float64)
        if T.size != 25 or E.size != 80:
            raise ValueError("HMM must have 25 transitions and 80 emissions")
        self._set(slot, (T, E))

    def stage(self, params, refs, reads, guides, model_slot=None, ref_index=None, guide_start=None):
        """refs/reads: lists of ASCII sequences (str/bytes); guides: list of [(op,len),...].
        ref_index[i] = which entry of `refs` read i aligns to (None: read i <-> refs[i]).
        guide_start[i] = (first reference position, first read position) of guide i -- the coordinates of the
        exonerate cigar line cactus_realign reads; None: every guide is global over both sequences."""
        ref, ref_off = _csr(refs)
        read, read_off = _csr(reads)
        gops, goff = _csr_ops(guides)
        ms = None if model_slot is None else np.ascontiguousarray(model_slot, dtype=np.int32)
        ri = None if ref_index is None else np.ascontiguousarray(ref_index, dtype=np.int32)
        gs = None if guide_start is None else np.ascontiguousarray(guide_start, dtype=np.int64).reshape(-1, 2)
        return Batch(self, params, ref, ref_off, read, read_off, gops, goff, ms, ri, gs)

    def stage_csr(self, params, ref, ref_off, read, read_off, guide_ops, guide_off, model_slot=None,
                  ref_index=None, guide_start=None):
        ref = np.ascontiguousarray(ref, dtype=np.uint8)
        read = np.ascontiguousarray(read, dtype=np.uint8)
        ref_off = np.ascontiguousarray(ref_off, dtype=np.int64)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        guide_ops = np.ascontiguousarray(guide_ops, dtype=np.int32).reshape(-1, 2)
        guide_off = np.ascontiguousarray(guide_off, dtype=np.int64)
        ms = None if model_slot is None else np.ascontiguousarray(model_slot, dtype=np.int32)
        ri = None if ref_index is None else np.ascontiguousarray(ref_index, dtype=np.int32)
        gs = None if guide_start is None else np.ascontiguousarray(guide_start, dtype=np.int64).reshape(-1, 2)
        return Batch(self, params, ref, ref_off, read, read_off, guide_ops, guide_off, ms, ri, gs)

    def stage_spans(self, params, ref, ref_off, text, read_begin, read_end, guide_ops, guide_off, model_slot=None,
                    ref_index=None, guide_start=None):
        """Stages reads that lie scattered in `text` (uint8): read i = text[read_begin[i] : read_end[i]]
        (include/nprealign.h: npr_batch_create_spans).  guide_off may be a slice of a longer CSR: its values index guide_ops."""
        ms = None if model_slot is None else np.ascontiguousarray(model_slot, dtype=np.int32)
        ri = None if ref_index is None else np.ascontiguousarray(ref_index, dtype=np.int32)
        gs = None if guide_start is None else np.ascontiguousarray(guide_start, dtype=np.int64).reshape(-1, 2)
        return Batch(self, params, ref, np.ascontiguousarray(ref_off, dtype=np.int64), text,
                     np.ascontiguousarray(read_begin, dtype=np.int64), np.ascontiguousarray(guide_ops, dtype=np.int32).reshape(-1, 2),
                     np.ascontiguousarray(guide_off, dtype=np.int64), ms, ri, gs, read_end=np.ascontiguousarray(read_end, dtype=np.int64))

    def align_stats(self, refs, reads, cigars, ref_index=None, start=None):
        """Per-read reductions over the aligned pairs of arbitrary alignments (a mapper's SAM records) on the device
        (include/nprealign.h: npr_align_stats).  refs / reads: ASCII sequences; cigars: [(op, len)] lists with ops M/I/D
        (0/1/2); start[i] = (first reference position, first read position) of cigar i.  int32 [n, STATS_WORDS]."""
        ref, ref_off = _csr(refs)
        read, read_off = _csr(reads)
        ops, ops_off = _csr_ops(cigars)
        n = len(read_off) - 1
        ri = None if ref_index is None else np.ascontiguousarray(ref_index, dtype=np.int32)
        st = None if start is None else np.ascontiguousarray(start, dtype=np.int64).reshape(-1, 2)
        out = np.zeros((n, _lib.STATS_WORDS), dtype=np.int32)
        rc = self._L.npr_align_stats(self._h, n, len(ref_off) - 1, ptr(ref), ptr(ref_off), ptr(ri), ptr(read), ptr(read_off),
                                     ptr(ops), ptr(ops_off), ptr(st), ptr(out))
        if rc != _lib.OK:
            raise NprError(rc, "npr_align_stats", self.last_error())
        return out

    def _realign_once(self, params, refs, reads, guides, model_slot, want_pairs, ref_index, guide_start=None):
        b = self.stage(params, refs, reads, guides, model_slot, ref_index, guide_start)
        try:
            b.run()
            b.finish()
            res = b.results()
            off, ops = b.ops()
            out = []
            aoff, arith = b.segment_arith()
            if want_pairs:
                poff, x, y, p = b.pairs()
            for i in range(b.n_reads):
                d = dict(status=int(res["status"][i]), score=float(res["score"][i]), loglik=float(res["loglik"][i]),
                         loglik_bwd=float(res["loglik_bwd"][i]), cells=int(res["cells"][i]),
                         n_segments=int(res["n_segments"][i]), n_pairs=int(res["n_pairs"][i]),
                         ops=[(int(a), int(c)) for a, c in ops[off[i]:off[i + 1]]],
                         seg_arith=[int(v) for v in arith[aoff[i]:aoff[i + 1]]])
                if want_pairs:
                    d["x"] = x[poff[i]:poff[i + 1]].copy()
                    d["y"] = y[poff[i]:poff[i + 1]].copy()
                    d["p"] = p[poff[i]:poff[i + 1]].copy()
                out.append(d)
            return out
        finally:
            b.close()

    def realign(self, params, refs, reads, guides, model_slot=None, want_pairs=False, ref_index=None, guide_start=None):
        """One batched call: returns list of dicts (status, score, loglik, cells, ops[, x, y, p]).

        Reads whose sparse posterior list overflowed its capacity (NPR_ERR_CAPACITY: a diffuse model can put up to
        1/threshold pairs on a base) are re-run with a four times larger `max_pairs_per_base` until they fit."""
        out = self._realign_once(params, refs, reads, guides, model_slot, want_pairs, ref_index, guide_start)
        per_base = params.max_pairs_per_base if params.max_pairs_per_base > 0 else 6
        limit = int(1.0 / max(params.posterior_threshold, 1e-6)) + 1
        while per_base < limit:
            again = [i for i, o in enumerate(out) if o["status"] == _lib.ERR_CAPACITY]
            if not again:
                break
            per_base = min(4 * per_base, limit)
            p2 = Params.from_buffer_copy(params)
            p2.max_pairs_per_base = per_base
            if ref_index is None:
                sub_refs, sub_index = [refs[i] for i in again], None
            else:
                sub_refs, sub_index = refs, [ref_index[i] for i in again]
            sub = self._realign_once(p2, sub_refs, [reads[i] for i in again], [guides[i] for i in again],
                                     None if model_slot is None else [model_slot[i] for i in again], want_pairs, sub_index,
                                     None if guide_start is None else [guide_start[i] for i in again])
            for i, o in zip(again, sub):
                out[i] = o
        return out

    def close(self):
        if getattr(self, "_h", None):
            # a batch that outlives its context (a test that failed with one open, at interpreter exit) would hand npr_batch_destroy
            # a batch whose context is gone: the library has no way to know
            for b in list(getattr(self, "_open", ())):
                b.close()
            self._L.npr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- host logic (no GPU needed) ----

def plan(params, lX, lY, guide):
    L = _lib.load()
    g = np.ascontiguousarray(np.asarray(guide, dtype=np.int32).reshape(-1, 2))
    h = C.c_void_p()
    rc = L.npr_plan_create(C.byref(params), lX, lY, ptr(g), len(g), C.byref(h))
    if rc != _lib.OK:
        raise NprError(rc, "npr_plan_create")
    out = []
    try:
        for s in range(L.npr_plan_segments(h)):
            info = np.zeros(8, dtype=np.int64)
            L.npr_plan_segment_info(h, s, ptr(info))
            D = int(info[6])
            lo = np.zeros(D + 1, dtype=np.int32)
            n = np.zeros(D + 1, dtype=np.int32)
            L.npr_plan_segment_band(h, s, ptr(lo), ptr(n))
            out.append(dict(xs=int(info[0]), ys=int(info[1]), xe=int(info[2]), ye=int(info[3]),
                            ragged_start=int(info[4]), ragged_end=int(info[5]), D=D, cells=int(info[7]), lo=lo,
                            n=n))
    finally:
        L.npr_plan_destroy(h)
    return out


def frame_schedule(params, lX, lY, guide, slots, slots_per_lane, segment=0):
    """Frame schedule of the register kernels for one segment of the plan (host logic, no GPU needed):
    dict(jlo, rebase, row_off, cells) or None when a frame of `slots` slots cannot follow the band
    (include/nprealign.h: npr_plan_frame_schedule)."""
    L = _lib.load()
    g = np.ascontiguousarray(np.asarray(guide, dtype=np.int32).reshape(-1, 2))
    h = C.c_void_p()
    rc = L.npr_plan_create(C.byref(params), lX, lY, ptr(g), len(g), C.byref(h))
    if rc != _lib.OK:
        raise NprError(rc, "npr_plan_create")
    try:
        info = np.zeros(8, dtype=np.int64)
        L.npr_plan_segment_info(h, segment, ptr(info))
        D = int(info[6])
        jlo = np.zeros(D + 1, dtype=np.int32)
        reb = np.zeros(D + 1, dtype=np.int32)
        off = np.zeros(D + 1, dtype=np.uint32)
        cells = np.zeros(1, dtype=np.int64)
        rc = L.npr_plan_frame_schedule(h, segment, slots, slots_per_lane, ptr(jlo), ptr(reb), ptr(off), ptr(cells))
        if rc == _lib.ERR_BAND_TOO_WIDE:
            return None
        if rc != _lib.OK:
            raise NprError(rc, "npr_plan_frame_schedule")
        return dict(jlo=jlo, rebase=reb, row_off=off, cells=int(cells[0]))
    finally:
        L.npr_plan_destroy(h)


def stripes(params, lX, lY, guide, slots_per_lane=2, segment=0):
    """Stripe table of the wide-band kernel for one segment of the plan (host logic, no GPU needed):
    dict(X, K, df, dl, row0 -- arrays per stripe --, rows) (include/nprealign.h: npr_plan_stripes)."""
    L = _lib.load()
    g = np.ascontiguousarray(np.asarray(guide, dtype=np.int32).reshape(-1, 2))
    h = C.c_void_p()
    rc = L.npr_plan_create(C.byref(params), lX, lY, ptr(g), len(g), C.byref(h))
    if rc != _lib.OK:
        raise NprError(rc, "npr_plan_create")
    try:
        S = L.npr_plan_stripes(h, segment, slots_per_lane, None, 0, None)
        if S < 0:
            raise NprError(S, "npr_plan_stripes")
        tab = np.zeros((S, 5), dtype=np.int32)
        rows = np.zeros(1, dtype=np.int64)
        rc = L.npr_plan_stripes(h, segment, slots_per_lane, ptr(tab), S, ptr(rows))
        if rc < 0:
            raise NprError(rc, "npr_plan_stripes")
        return dict(X=tab[:, 0].copy(), K=tab[:, 1].copy(), df=tab[:, 2].copy(), dl=tab[:, 3].copy(),
                    row0=tab[:, 4].copy(), rows=int(rows[0]))
    finally:
        L.npr_plan_destroy(h)


def format_cigars(ops_off, ops):
    """SAM CIGAR text of every op list of a CSR pair as Batch.ops() returns it: (bytes buffer, offsets[n+1])
    (include/nprealign.h: npr_format_cigars)."""
    L = _lib.load()
    ops_off = np.ascontiguousarray(ops_off, dtype=np.int64)
    ops = np.ascontiguousarray(ops, dtype=np.int32).reshape(-1, 2)
    n = len(ops_off) - 1
    str_off = np.zeros(n + 1, dtype=np.int64)
    total = L.npr_format_cigars(n, ptr(ops_off), ptr(ops), ptr(str_off), None, 0)
    if total < 0:
        raise NprError(int(total), "npr_format_cigars")
    buf = np.empty(max(int(total), 1), dtype=np.uint8)
    rc = L.npr_format_cigars(n, ptr(ops_off), ptr(ops), ptr(str_off), ptr(buf), int(total))
    if rc < 0:
        raise NprError(int(rc), "npr_format_cigars")
    return buf[:int(total)], str_off


def format_cigars_packed(word_off, n_ops, words):
    """SAM CIGAR text from packed cigars (one uint32 per op, length << 2 | op), list i at words[word_off[i] .. + n_ops[i]):
    (bytes buffer, offsets[n+1]) (include/nprealign.h: npr_format_cigars_packed)."""
    L = _lib.load()
    word_off = np.ascontiguousarray(word_off, dtype=np.int64)
    n_ops = np.ascontiguousarray(n_ops, dtype=np.int64)
    words = np.ascontiguousarray(words, dtype=np.uint32)
    n = len(n_ops)
    str_off = np.zeros(n + 1, dtype=np.int64)
    total = L.npr_format_cigars_packed(n, ptr(word_off), ptr(n_ops), ptr(words), ptr(str_off), None, 0)
    if total < 0:
        raise NprError(int(total), "npr_format_cigars_packed")
    buf = np.empty(max(int(total), 1), dtype=np.uint8)
    rc = L.npr_format_cigars_packed(n, ptr(word_off), ptr(n_ops), ptr(words), ptr(str_off), ptr(buf), int(total))
    if rc < 0:
        raise NprError(int(rc), "npr_format_cigars_packed")
    return buf[:int(total)], str_off


def _ragged(items):
    """list of bytes -> (uint8 array, int64 offsets[n+1])"""
    off = np.zeros(len(items) + 1, dtype=np.int64)
    if len(items):
        np.cumsum(np.fromiter(map(len, items), dtype=np.int64, count=len(items)), out=off[1:])
    return np.frombuffer(b"".join(items) or b"\0", dtype=np.uint8), off


def format_sam_records(qnames, ref_names, ref_index, pos, word_off, n_ops, words, seq, seq_off, flag=None, mapq=None):
    """The realigned SAM records as one uint8 array + offsets[n+1] (include/nprealign.h: npr_format_sam_records).  qnames /
    ref_names: lists of bytes; pos 1-based; cigars packed as for format_cigars_packed; seq: uint8 array of read bases with
    CSR offsets seq_off (relative to seq[0])."""
    L = _lib.load()
    n = len(qnames)
    qn, qoff = _ragged(qnames)
    rn, roff = _ragged(ref_names)
    ref_index = np.ascontiguousarray(ref_index, dtype=np.int32)
    pos = np.ascontiguousarray(pos, dtype=np.int64)
    word_off = np.ascontiguousarray(word_off, dtype=np.int64)
    n_ops = np.ascontiguousarray(n_ops, dtype=np.int64)
    words = np.ascontiguousarray(words, dtype=np.uint32)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
    flag = None if flag is None else np.ascontiguousarray(flag, dtype=np.int32)
    mapq = None if mapq is None else np.ascontiguousarray(mapq, dtype=np.int32)
    rec_off = np.zeros(n + 1, dtype=np.int64)
    args = [n, ptr(qn), ptr(qoff), None if flag is None else ptr(flag), ptr(rn), ptr(roff), ptr(ref_index), ptr(pos),
            None if mapq is None else ptr(mapq), ptr(word_off), ptr(n_ops), ptr(words), ptr(seq), ptr(seq_off), ptr(rec_off)]
    total = L.npr_format_sam_records(*args, None, 0)
    if total < 0:
        raise NprError(int(total), "npr_format_sam_records")
    buf = np.empty(max(int(total), 1), dtype=np.uint8)
    rc = L.npr_format_sam_records(*args, ptr(buf), int(total))
    if rc < 0:
        raise NprError(int(rc), "npr_format_sam_records")
    return buf[:int(total)], rec_off


def mea_cigar(lX, lY, x, y, p, gap_gamma=0.5, match_gamma=0.0):
    L = _lib.load()
    x = np.ascontiguousarray(x, dtype=np.int32)
    y = np.ascontiguousarray(y, dtype=np.int32)
    p = np.ascontiguousarray(p, dtype=np.float32)
    cap = 2 * len(x) + 8
    ops = np.zeros((cap, 2), dtype=np.int32)
    score = C.c_double(0)
    k = L.npr_mea_cigar(lX, lY, ptr(x), ptr(y), ptr(p), len(x), gap_gamma, match_gamma, ptr(ops), cap,
                        C.byref(score))
    if k < 0:
        raise NprError(int(k), "npr_mea_cigar")
    return [(int(a), int(c)) for a, c in ops[:k]], score.value


def rescore(guide, x, y, p):
    L = _lib.load()
    g = np.ascontiguousarray(np.asarray(guide, dtype=np.int32).reshape(-1, 2))
    x = np.ascontiguousarray(x, dtype=np.int32)
    y = np.ascontiguousarray(y, dtype=np.int32)
    p = np.ascontiguousarray(p, dtype=np.float32)
    score = C.c_double(0)
    rc = L.npr_rescore(ptr(g), len(g), ptr(x), ptr(y), ptr(p), len(x), C.byref(score))
    if rc != _lib.OK:
        raise NprError(rc, "npr_rescore")
    return score.value


def encode(seq):
    L = _lib.load()
    if isinstance(seq, str):
        seq = seq.encode("ascii")
    a = np.frombuffer(seq, dtype=np.uint8)
    out = np.zeros(len(a), dtype=np.uint8)
    L.npr_encode_bases(ptr(a), len(a), ptr(out))
    return out
