"""ctypes binding of libnprealign.so (C ABI: include/nprealign.h).

The library is the only execution path of the realigner: if it is missing this module raises
ImportError-like RuntimeError on first use, and `Context()` raises when no gfx950 device is usable.
There is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_LOADED = False  # set when the library (and with it the HIP runtime) has been loaded into this process


def want_hw_queues(n=8):
    """The files -> file job (job.py) keeps three batches in flight, each with its own streams; the HIP runtime spreads a process's streams over
    GPU_MAX_HW_QUEUES hardware queues (default 4) and a queue runs its kernels in submission order, so with four the MEA stage of one chunk
    and the DP pass of the next regularly share one (403 against 393 ms per 50 000 reads).  The runtime reads the variable
    at the process's first HIP call.  This is a DEPLOYMENT setting: importing the binding changes nothing in the host
    application's environment -- the job's entry points and bench.py call this, it leaves a value the host has chosen alone, and it says on
    stderr (NPR_TIMING / NPR_JOB_TRACE) what it did, including when it comes too late to matter."""
    have = os.environ.get("GPU_MAX_HW_QUEUES")
    if have is None:
        os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    if os.environ.get("NPR_TIMING") or os.environ.get("NPR_JOB_TRACE"):
        import sys
        sys.stderr.write("[npr] GPU_MAX_HW_QUEUES %s%s\n" % (
            "left at the host's %s" % have if have is not None else "set to %d for this process and its children" % n,
            " (the library is already loaded: the runtime may have read it before)" if LIB_LOADED and have is None else ""))


LIB_PATH = os.environ.get("NPR_LIB") or os.path.join(_HERE, "libnprealign.so")  # NPR_LIB: a variant build (bring-up, tools/variant_bench.py)

OK = 0
ERR_INVALID, ERR_ZERO_PROB, ERR_CAPACITY, ERR_MODEL, ERR_NO_DEVICE, ERR_HIP, ERR_BAND_TOO_WIDE, ERR_NOMEM, \
    ERR_STATE = -1, -2, -3, -4, -5, -6, -7, -8, -9
OP_M, OP_I, OP_D = 0, 1, 2
BAND_ANCHOR, BAND_FIXED = 0, 1
MODE_REALIGN, MODE_RESCORE_ORIGINAL, MODE_ALL_POSTERIORS, MODE_EXPECTATIONS = 0, 1, 2, 3
OPT_OVERLAP = 1
OPT_RELEASE_SCRATCH = 2
# test / bring-up switches (include/nprealign.h: NPR_OPT_*; none changes a result)
OPTIONS = dict(kernel=3, arith=4, pair=5, no_tile=6, no_wide=7, tile_rs=8, tile_waves=9, waves_per_cu=10, class_min=11,
               variable_scratch=12, host_mea=13, mea_ring_only=14, mea_global_sort=15, mea_own_scratch=16, em_generic=17,
               em_serial=18, em_waves=19, mea_wide_ops=20, em_tile=21)
KERNEL_GENERIC, ARITH_CELL, PAIR_NEVER, PAIR_LONG, PAIR_ALL = 1, 1, 1, 2, 3
STATS_WORDS = 40  # NPR_STATS_WORDS
MAX_MODELS = 8
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
        ("max_pairs_per_base", C.c_int32),
    ]


class ReadResult(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("n_segments", C.c_int32),
        ("cells", C.c_int64),
        ("loglik", C.c_double),
        ("loglik_bwd", C.c_double),
        ("score", C.c_double),
        ("n_ops", C.c_int64),
        ("n_pairs", C.c_int64),
    ]


RESULT_DTYPE = np.dtype([("status", np.int32), ("n_segments", np.int32), ("cells", np.int64),
                         ("loglik", np.float64), ("loglik_bwd", np.float64), ("score", np.float64),
                         ("n_ops", np.int64), ("n_pairs", np.int64)], align=True)


class BatchStats(C.Structure):
    _fields_ = [
        ("n_reads", C.c_int64),
        ("n_tasks", C.c_int64),
        ("cells", C.c_int64),
        ("diagonals", C.c_int64),
        ("max_width", C.c_int64),
        ("device_bytes", C.c_int64),
        ("slots", C.c_int64),
        ("kernel_variant", C.c_int32),
    ]


class NprError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        msg = "%s failed: %s (%d)" % (where, strerror(code), code)
        if detail:
            msg += " -- " + detail
        RuntimeError.__init__(self, msg)


# every symbol include/nprealign.h declares
EXPORTS = [
    "npr_abi_version", "npr_strerror", "npr_create", "npr_destroy", "npr_last_error", "npr_set_hmm",
    "npr_batch_create", "npr_batch_create_at", "npr_batch_run", "npr_batch_finish", "npr_batch_destroy", "npr_batch_get_stats", "npr_batch_class_stats",
    "npr_batch_results", "npr_batch_ops", "npr_batch_ops_packed", "npr_batch_pairs", "npr_batch_debug_set_pairs", "npr_batch_dense", "npr_batch_rs_forward", "npr_batch_expectations",
    "npr_batch_align_stats", "npr_align_stats", "npr_batch_plan_check", "npr_batch_base_expectations",
    "npr_realign_batch",
    "npr_plan_create", "npr_plan_destroy", "npr_plan_segments", "npr_plan_segment_info",
    "npr_plan_segment_band", "npr_plan_frame_schedule", "npr_plan_stripes", "npr_format_cigars", "npr_format_cigars_packed", "npr_format_sam_records", "npr_chain_hits", "npr_mea_cigar", "npr_rescore", "npr_encode_bases",
    "npr_sam_index", "npr_sam_parse", "npr_sam_guides", "npr_sam_splice", "npr_fasta_index", "npr_fasta_pack", "npr_fastq_index",
    "npr_batch_create_spans", "npr_chain_merge", "npr_ctx_option", "npr_batch_segment_arith",
]

_lib = None


def load():
    """Loads libnprealign.so; raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libnprealign.so not found at %s: build it with `python -c 'import __graft_entry__ as g; "
                           "g.build()'` or `make -C nanopore_amd/csrc`; there is no CPU fallback" % LIB_PATH)
    global LIB_LOADED
    L = C.CDLL(LIB_PATH)
    LIB_LOADED = True
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.npr_abi_version.restype = i32
    L.npr_strerror.restype = C.c_char_p
    L.npr_strerror.argtypes = [i32]
    L.npr_create.restype = i32
    L.npr_create.argtypes = [i32, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.npr_destroy.restype = None
    L.npr_destroy.argtypes = [vp]
    L.npr_last_error.restype = C.c_char_p
    L.npr_last_error.argtypes = [vp]
    L.npr_ctx_option.restype = i32
    L.npr_ctx_option.argtypes = [vp, i32, i64]
    L.npr_set_hmm.restype = i32
    L.npr_set_hmm.argtypes = [vp, i32, vp, vp]
    L.npr_batch_create.restype = i32
    L.npr_batch_create.argtypes = [vp, C.POINTER(Params), i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(vp)]
    L.npr_plan_frame_schedule.restype = i32
    L.npr_plan_frame_schedule.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
    L.npr_batch_class_stats.restype = i32
    L.npr_batch_class_stats.argtypes = [vp, vp, vp, i32]
    L.npr_batch_segment_arith.restype = i32
    L.npr_batch_segment_arith.argtypes = [vp, vp, vp, i64]
    L.npr_batch_create_at.restype = i32
    L.npr_batch_create_at.argtypes = [vp, C.POINTER(Params), i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(vp)]
    L.npr_batch_run.restype = i32
    L.npr_batch_run.argtypes = [vp, C.POINTER(C.c_float)]
    L.npr_batch_finish.restype = i32
    L.npr_batch_finish.argtypes = [vp]
    L.npr_batch_destroy.restype = None
    L.npr_batch_destroy.argtypes = [vp]
    L.npr_batch_get_stats.restype = i32
    L.npr_batch_get_stats.argtypes = [vp, C.POINTER(BatchStats)]
    L.npr_batch_results.restype = i32
    L.npr_batch_results.argtypes = [vp, vp]
    L.npr_batch_ops.restype = i32
    L.npr_batch_ops.argtypes = [vp, vp, vp, i64]
    L.npr_batch_ops_packed.restype = i32
    L.npr_batch_ops_packed.argtypes = [vp, vp, vp, i64]
    L.npr_batch_pairs.restype = i32
    L.npr_batch_pairs.argtypes = [vp, vp, vp, vp, vp, i64]
    L.npr_batch_debug_set_pairs.restype = i32
    L.npr_batch_debug_set_pairs.argtypes = [vp, i64, vp, vp, vp, i64, i32]
    L.npr_batch_expectations.restype = i32
    L.npr_batch_expectations.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_float)]
    L.npr_batch_base_expectations.restype = i32
    L.npr_batch_base_expectations.argtypes = [vp, vp, i64, vp, vp, vp]
    L.npr_batch_plan_check.restype = i64
    L.npr_batch_plan_check.argtypes = [vp, vp]
    L.npr_batch_align_stats.restype = i32
    L.npr_batch_align_stats.argtypes = [vp, vp]
    L.npr_align_stats.restype = i32
    L.npr_align_stats.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.npr_batch_dense.restype = i32
    L.npr_batch_dense.argtypes = [vp, i64, vp, vp, vp, vp, i64]
    L.npr_batch_rs_forward.restype = i32
    L.npr_batch_rs_forward.argtypes = [vp, i64, vp, vp, i64]
    L.npr_realign_batch.restype = i32
    L.npr_realign_batch.argtypes = [vp, C.POINTER(Params), i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64]
    L.npr_plan_create.restype = i32
    L.npr_plan_create.argtypes = [C.POINTER(Params), i64, i64, vp, i64, C.POINTER(vp)]
    L.npr_plan_destroy.restype = None
    L.npr_plan_destroy.argtypes = [vp]
    L.npr_plan_segments.restype = i32
    L.npr_plan_segments.argtypes = [vp]
    L.npr_plan_segment_info.restype = i32
    L.npr_plan_segment_info.argtypes = [vp, i32, vp]
    L.npr_plan_segment_band.restype = i32
    L.npr_plan_segment_band.argtypes = [vp, i32, vp, vp]
    L.npr_plan_stripes.restype = i32
    L.npr_plan_stripes.argtypes = [vp, i32, i32, vp, i32, vp]
    L.npr_format_cigars_packed.restype = i64
    L.npr_format_cigars_packed.argtypes = [i64, vp, vp, vp, vp, vp, i64]
    L.npr_format_sam_records.restype = i64
    L.npr_format_sam_records.argtypes = [i64] + [vp] * 15 + [i64]
    L.npr_chain_hits.restype = i64
    L.npr_chain_hits.argtypes = [i64, vp, vp, vp, vp, vp, vp, i64, vp]
    L.npr_chain_merge.restype = i64
    L.npr_chain_merge.argtypes = [i64, vp, vp, vp, vp, i64, i64, vp, i64]
    L.npr_format_cigars.restype = i64
    L.npr_format_cigars.argtypes = [i64, vp, vp, vp, vp, i64]
    L.npr_mea_cigar.restype = i64
    L.npr_mea_cigar.argtypes = [i64, i64, vp, vp, vp, i64, dbl, dbl, vp, i64, C.POINTER(dbl)]
    L.npr_rescore.restype = i32
    L.npr_rescore.argtypes = [vp, i64, vp, vp, vp, i64, C.POINTER(dbl)]
    L.npr_encode_bases.restype = None
    L.npr_encode_bases.argtypes = [vp, i64, vp]
    L.npr_sam_index.restype = i64
    L.npr_sam_index.argtypes = [vp, i64, C.POINTER(i64), vp, i64]
    L.npr_sam_parse.restype = i32
    L.npr_sam_parse.argtypes = [vp, vp, i64, vp, vp, i64, vp]
    L.npr_sam_guides.restype = i32
    L.npr_sam_guides.argtypes = [vp, vp, i64, vp, vp]
    L.npr_sam_splice.restype = i64
    L.npr_sam_splice.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, vp, i64]
    L.npr_fasta_index.restype = i64
    L.npr_fasta_index.argtypes = [vp, i64, vp, vp, i64]
    L.npr_fasta_pack.restype = i32
    L.npr_fasta_pack.argtypes = [vp, vp, i64, vp, vp]
    L.npr_fastq_index.restype = i64
    L.npr_fastq_index.argtypes = [vp, i64, vp, i64]
    L.npr_batch_create_spans.restype = i32
    L.npr_batch_create_spans.argtypes = [vp, C.POINTER(Params), i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(vp)]
    _lib = L
    return L


def strerror(code):
    return load().npr_strerror(code).decode()


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
