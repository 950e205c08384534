"""ctypes wrapper around the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package (nanopore_amd).  See oracle/realign_oracle.h for the parity
statement (PARITY UNPINNED: the reference's realigner, cactus_realign, called at
nanopore/analyses/utils.py:587, is absent from the snapshot).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

BAND_ANCHOR, BAND_FIXED = 0, 1
MODE_REALIGN, MODE_RESCORE_ORIGINAL, MODE_ALL_POSTERIORS = 0, 1, 2
OP_M, OP_I, OP_D = 0, 1, 2
E_DEAD = -(1 << 28)


class Params(C.Structure):
    _fields_ = [
        ("band_mode", C.c_int32),
        ("diagonal_expansion", C.c_int32),
        ("constraint_trim", C.c_int32),
        ("split_threshold", C.c_int64),
        ("fixed_width", C.c_int32),
        ("gap_gamma", C.c_double),
        ("match_gamma", C.c_double),
        ("posterior_threshold", C.c_double),
        ("mode", C.c_int32),
    ]


def make_params(band_mode=BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=3000,
                fixed_width=0, gap_gamma=0.5, match_gamma=0.0, posterior_threshold=0.01, mode=MODE_REALIGN):
    return Params(band_mode, diagonal_expansion, constraint_trim, split_threshold, fixed_width, gap_gamma,
                  match_gamma, posterior_threshold, mode)


class Hmm(C.Structure):
    _fields_ = [("T", C.c_double * 25), ("E", C.c_double * 80)]


class ReadResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("cells", C.c_int64), ("total_ll", C.c_double), ("score", C.c_double),
                ("nops", C.c_int64), ("npairs", C.c_int64)]


def build(native=False, force=False):
    """Compile the oracle.  native=True builds liboracle_native.so with -march=native (CPU baseline)."""
    name = "liboracle_native.so" if native else "liboracle.so"
    path = os.path.join(_HERE, name)
    srcs = [os.path.join(_HERE, f) for f in ("realign_oracle.c", "realign_oracle_f32.c", "realign_oracle_rs.c", "realign_oracle.h")]
    if not force and os.path.exists(path) and all(os.path.getmtime(path) >= os.path.getmtime(s) for s in srcs):
        return path
    march = "-march=native" if native else "-march=x86-64-v2"
    cmd = ["gcc", "-O3", march, "-std=gnu11", "-fPIC", "-fopenmp", "-ffp-contract=off", "-shared", "-o", path,
           srcs[0], srcs[1], srcs[2], "-lm"]
    subprocess.check_call(cmd)
    return path


_libs = {}


def lib(native=False):
    if native not in _libs:
        path = build(native=native)
        L = C.CDLL(path)
        L.orc_plan_build.restype = C.c_void_p
        L.orc_plan_build.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(Params),
                                     C.POINTER(C.c_int32)]
        L.orc_plan_free.argtypes = [C.c_void_p]
        L.orc_plan_nseg.argtypes = [C.c_void_p]
        L.orc_plan_seg_info.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.orc_plan_seg_band.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.orc_fb_f64.restype = C.c_int32
        L.orc_fb_f64.argtypes = [C.POINTER(Hmm), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                 C.c_void_p, C.c_int32, C.c_int32, C.c_double] + [C.c_void_p] * 9 + [
                                     C.c_int64, C.c_void_p]
        L.orc_fb_f32.restype = C.c_int32
        L.orc_fb_f32.argtypes = [C.POINTER(Hmm), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                 C.c_void_p, C.c_int32, C.c_int32, C.c_float] + [C.c_void_p] * 11 + [
                                     C.c_int64, C.c_void_p]
        L.orc_fb_f32_rs.restype = C.c_int32
        L.orc_fb_f32_rs.argtypes = L.orc_fb_f32.argtypes
        L.orc_expectations_f64.restype = C.c_int32
        L.orc_expectations_f64.argtypes = [C.POINTER(Hmm), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                           C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mea_cigar.restype = C.c_int64
        L.orc_mea_cigar.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_double, C.c_double, C.c_void_p, C.c_int64, C.POINTER(C.c_double),
                                    C.c_int32]
        L.orc_rescore.restype = C.c_double
        L.orc_rescore.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_realign_read.restype = C.c_int32
        L.orc_realign_read.argtypes = [C.POINTER(Hmm), C.POINTER(Params), C.c_int32, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(ReadResult)]
        L.orc_realign_read_arith.restype = C.c_int32
        L.orc_realign_read_arith.argtypes = [C.POINTER(Hmm), C.POINTER(Params), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(ReadResult)]
        L.orc_realign_batch_arith.restype = C.c_int32
        L.orc_realign_batch_arith.argtypes = [C.POINTER(Hmm), C.POINTER(Params), C.c_int32, C.c_void_p, C.c_void_p, C.c_int64] + [
            C.c_void_p] * 13 + [C.c_int32]
        L.orc_realign_batch.restype = C.c_int32
        L.orc_realign_batch.argtypes = [C.POINTER(Hmm), C.POINTER(Params), C.c_int32, C.c_int64] + [
            C.c_void_p] * 13 + [C.c_int32]
        L.orc_set_logadd_kind.argtypes = [C.c_int32]
        L.orc_set_logadd_kind.restype = None
        L.orc_get_logadd_kind.restype = C.c_int32
        L.orc_logadd.restype = C.c_double
        L.orc_logadd.argtypes = [C.c_double, C.c_double]
        _libs[native] = L
    return _libs[native]


LOGADD_EXACT, LOGADD_APPROX = 0, 1


class logadd_kind(object):
    """with logadd_kind(LOGADD_APPROX): ... -- the fp64 passes use cPecan's piecewise-cubic log-add [RECALLED, SURVEY.md
    Appendix A] inside the block (both builds of the library), the exact one again afterwards."""

    def __init__(self, kind):
        self.kind = kind

    def __enter__(self):
        for native in (False, True):
            lib(native).orc_set_logadd_kind(self.kind)
        return self

    def __exit__(self, *exc):
        for L in _libs.values():
            L.orc_set_logadd_kind(LOGADD_EXACT)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make_hmm(T, E):
    h = Hmm()
    T = np.asarray(T, dtype=np.float64).reshape(25)
    E = np.asarray(E, dtype=np.float64).reshape(80)
    for i in range(25):
        h.T[i] = T[i]
    for i in range(80):
        h.E[i] = E[i]
    return h


def ops_array(ops):
    """[(op,len),...] -> contiguous int32 array of pairs."""
    a = np.ascontiguousarray(np.asarray(ops, dtype=np.int32).reshape(-1, 2))
    return a


def plan(lX, lY, ops, params):
    """Returns list of segments: dict(xs,ys,xe,ye,ragged_start,ragged_end,D,cells,lo,n)."""
    L = lib()
    ops = ops_array(ops)
    st = C.c_int32(0)
    h = L.orc_plan_build(lX, lY, _p(ops), len(ops), C.byref(params), C.byref(st))
    if not h:
        raise ValueError("orc_plan_build failed: status %d" % st.value)
    out = []
    try:
        for s in range(L.orc_plan_nseg(h)):
            info = np.zeros(8, dtype=np.int64)
            L.orc_plan_seg_info(h, s, _p(info))
            D = int(info[6])
            lo = np.zeros(D + 1, dtype=np.int32)
            n = np.zeros(D + 1, dtype=np.int32)
            L.orc_plan_seg_band(h, s, _p(lo), _p(n))
            out.append(dict(xs=int(info[0]), ys=int(info[1]), xe=int(info[2]), ye=int(info[3]),
                            ragged_start=int(info[4]), ragged_end=int(info[5]), D=D, cells=int(info[7]),
                            lo=lo, n=n))
    finally:
        L.orc_plan_free(h)
    return out


def fb_f64(hmm, X, Y, lo, n, ragged_start=0, ragged_end=0, threshold=0.01, dense=True, all_states=False):
    L = lib()
    X = np.ascontiguousarray(X, dtype=np.uint8)
    Y = np.ascontiguousarray(Y, dtype=np.uint8)
    lo = np.ascontiguousarray(lo, dtype=np.int32)
    n = np.ascontiguousarray(n, dtype=np.int32)
    cells = int(n.sum())
    Fm = np.zeros(cells) if dense else None
    Bm = np.zeros(cells) if dense else None
    Fall = np.zeros((cells, 5)) if all_states else None
    Ball = np.zeros((cells, 5)) if all_states else None
    cap = 4 * cells + 16
    px = np.zeros(cap, dtype=np.int32)
    py = np.zeros(cap, dtype=np.int32)
    pp = np.zeros(cap, dtype=np.float64)
    tot = C.c_double(0)
    totb = C.c_double(0)
    npairs = C.c_int64(0)
    rc = L.orc_fb_f64(C.byref(hmm), _p(X), len(X), _p(Y), len(Y), _p(lo), _p(n), ragged_start, ragged_end,
                      threshold, C.addressof(tot), C.addressof(totb), _p(Fm), _p(Bm), _p(Fall), _p(Ball),
                      _p(px), _p(py), _p(pp), cap, C.addressof(npairs))
    k = npairs.value
    return dict(rc=rc, total_ll=tot.value, total_ll_bwd=totb.value, Fm=Fm, Bm=Bm, Fall=Fall, Ball=Ball,
                px=px[:k].copy(), py=py[:k].copy(), pp=pp[:k].copy())


def fb_f32(hmm, X, Y, lo, n, ragged_start=0, ragged_end=0, threshold=0.01, dense=True, arith=0):
    """fp32 mirror of the device arithmetic: arith 0 = one exponent per cell (realign_oracle_f32.c), 1 = one per anti-diagonal
    row (realign_oracle_rs.c)."""
    L = lib()
    X = np.ascontiguousarray(X, dtype=np.uint8)
    Y = np.ascontiguousarray(Y, dtype=np.uint8)
    lo = np.ascontiguousarray(lo, dtype=np.int32)
    n = np.ascontiguousarray(n, dtype=np.int32)
    cells = int(n.sum())
    Fv = np.zeros(cells, dtype=np.float32) if dense else None
    Fe = np.zeros(cells, dtype=np.int32) if dense else None
    Bv = np.zeros(cells, dtype=np.float32) if dense else None
    Be = np.zeros(cells, dtype=np.int32) if dense else None
    cap = 4 * cells + 16
    px = np.zeros(cap, dtype=np.int32)
    py = np.zeros(cap, dtype=np.int32)
    pp = np.zeros(cap, dtype=np.float32)
    tm, bm = C.c_float(0), C.c_float(0)
    te, be = C.c_int32(0), C.c_int32(0)
    npairs = C.c_int64(0)
    rc = (L.orc_fb_f32_rs if arith else L.orc_fb_f32)(C.byref(hmm), _p(X), len(X), _p(Y), len(Y), _p(lo), _p(n), ragged_start, ragged_end,
                      threshold, C.addressof(tm), C.addressof(te), C.addressof(bm), C.addressof(be), _p(Fv),
                      _p(Fe), _p(Bv), _p(Be), _p(px), _p(py), _p(pp), cap, C.addressof(npairs))
    k = npairs.value
    return dict(rc=rc, tot_m=tm.value, tot_e=te.value, btot_m=bm.value, btot_e=be.value, Fm_v=Fv, Fm_e=Fe,
                Bm_v=Bv, Bm_e=Be, px=px[:k].copy(), py=py[:k].copy(), pp=pp[:k].copy())


def expectations(hmm, X, Y, lo, n, ragged_start=0, ragged_end=0):
    """Baum-Welch expected transition (25) and emission (80) counts of one banded segment, and its log-likelihood."""
    L = lib()
    X = np.ascontiguousarray(X, dtype=np.uint8)
    Y = np.ascontiguousarray(Y, dtype=np.uint8)
    lo = np.ascontiguousarray(lo, dtype=np.int32)
    n = np.ascontiguousarray(n, dtype=np.int32)
    T = np.zeros(25)
    E = np.zeros(80)
    ll = C.c_double(0)
    rc = L.orc_expectations_f64(C.byref(hmm), _p(X), len(X), _p(Y), len(Y), _p(lo), _p(n), ragged_start, ragged_end,
                                _p(T), _p(E), C.addressof(ll))
    return dict(rc=rc, T=T, E=E, total_ll=ll.value)


def mea_cigar(lX, lY, px, py, pp, gap_gamma=0.5, match_gamma=0.0, brute_force=False):
    L = lib()
    px = np.ascontiguousarray(px, dtype=np.int32)
    py = np.ascontiguousarray(py, dtype=np.int32)
    pp = np.ascontiguousarray(pp, dtype=np.float64)
    cap = 2 * (len(px) + 2) + 4
    out = np.zeros((cap, 2), dtype=np.int32)
    score = C.c_double(0)
    k = L.orc_mea_cigar(lX, lY, _p(px), _p(py), _p(pp), len(px), gap_gamma, match_gamma, _p(out), cap,
                        C.byref(score), 1 if brute_force else 0)
    if k < 0:
        raise RuntimeError("orc_mea_cigar failed: %d" % k)
    return [(int(a), int(b)) for a, b in out[:k]], score.value


def rescore(ops, px, py, pp):
    L = lib()
    ops = ops_array(ops)
    px = np.ascontiguousarray(px, dtype=np.int32)
    py = np.ascontiguousarray(py, dtype=np.int32)
    pp = np.ascontiguousarray(pp, dtype=np.float64)
    return L.orc_rescore(_p(ops), len(ops), _p(px), _p(py), _p(pp), len(px))


def realign_read(hmm, params, X, Y, guide_ops, precision=0, want_pairs=True, seg_arith=None):
    """seg_arith (precision 1 only): the fp32 arithmetic of each segment as the product reports it (`seg_arith` of a
    Context.realign result / Batch.segment_arith): 0 per-cell exponents, 1 row-scaled; None: 0 everywhere."""
    L = lib()
    sa = None if seg_arith is None else np.ascontiguousarray(seg_arith, dtype=np.int32)
    X = np.ascontiguousarray(X, dtype=np.uint8)
    Y = np.ascontiguousarray(Y, dtype=np.uint8)
    g = ops_array(guide_ops)
    cap_ops = 2 * (len(X) + len(Y)) + 8
    out = np.zeros((cap_ops, 2), dtype=np.int32)
    cap_pairs = 8 * min(len(X), len(Y)) + 4096
    px = np.zeros(cap_pairs, dtype=np.int32)
    py = np.zeros(cap_pairs, dtype=np.int32)
    pp = np.zeros(cap_pairs, dtype=np.float64)
    res = ReadResult()
    rc = L.orc_realign_read_arith(C.byref(hmm), C.byref(params), precision, _p(sa), 0 if sa is None else len(sa), _p(X), len(X),
                                  _p(Y), len(Y), _p(g), len(g), _p(out), cap_ops, _p(px), _p(py), _p(pp), cap_pairs, C.byref(res))
    k = res.npairs
    return dict(status=rc, cells=res.cells, total_ll=res.total_ll, score=res.score,
                ops=[(int(a), int(b)) for a, b in out[:res.nops]], px=px[:k].copy(), py=py[:k].copy(),
                pp=pp[:k].copy())


def realign_batch(hmm, params, X, x_off, Y, y_off, guide_ops, g_off, precision=0, threads=0, native=False, seg_arith=None):
    """CSR batch; returns dict(ops=list of arrays, score, total_ll, cells, status).  seg_arith: (seg_off, arith) as
    Batch.segment_arith returns them (precision 1: which fp32 arithmetic each segment ran in on the device)."""
    L = lib(native=native)
    X = np.ascontiguousarray(X, dtype=np.uint8)
    Y = np.ascontiguousarray(Y, dtype=np.uint8)
    x_off = np.ascontiguousarray(x_off, dtype=np.int64)
    y_off = np.ascontiguousarray(y_off, dtype=np.int64)
    g = np.ascontiguousarray(guide_ops, dtype=np.int32).reshape(-1, 2)
    g_off = np.ascontiguousarray(g_off, dtype=np.int64)
    nreads = len(x_off) - 1
    caps = 2 * ((x_off[1:] - x_off[:-1]) + (y_off[1:] - y_off[:-1])) + 8
    o_off = np.zeros(nreads + 1, dtype=np.int64)
    np.cumsum(caps, out=o_off[1:])
    out = np.zeros((int(o_off[-1]), 2), dtype=np.int32)
    nops = np.zeros(nreads, dtype=np.int64)
    score = np.zeros(nreads)
    ll = np.zeros(nreads)
    cells = np.zeros(nreads, dtype=np.int64)
    status = np.zeros(nreads, dtype=np.int32)
    so = sa = None
    if seg_arith is not None:
        so = np.ascontiguousarray(seg_arith[0], dtype=np.int64)
        sa = np.ascontiguousarray(seg_arith[1], dtype=np.int32)
        assert len(so) == nreads + 1
    L.orc_realign_batch_arith(C.byref(hmm), C.byref(params), precision, _p(so), _p(sa), nreads, _p(X), _p(x_off), _p(Y), _p(y_off),
                              _p(g), _p(g_off), _p(out), _p(o_off), _p(nops), _p(score), _p(ll), _p(cells),
                              _p(status), threads)
    ops = [out[o_off[i]:o_off[i] + nops[i]].copy() for i in range(nreads)]
    return dict(ops=ops, score=score, total_ll=ll, cells=cells, status=status)
