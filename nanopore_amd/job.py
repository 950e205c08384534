"""The whole realignment job on one read set, sharded over the ranks of a node.

The reference's job is: one jobTree job per record of ONE SAM file (nanopore/analyses/utils.py:565-570), the temp cigar
files gathered in input order and spliced into a copy of the input SAM (utils.py:591-609).  Here: every rank takes its
shard of the reads (`dist.shard_indices`: length-sorted, dealt round-robin), stages / realigns / closes it on its GPU with
no data-path collective, one chunked gather (RCCL under backend nccl) brings the packed cigars and scores to rank 0, which
restores the input order and writes the realigned SAM and a summary XML.  Used by `bench.py --workload c3` (strong
scaling, BASELINE.json configs[3]) and by the two-rank GPU test.
"""
import os
import time
import xml.etree.ElementTree as ET

import numpy as np

from . import dist as npd
from . import synth


def realign_shard(ctx, params, w, idx, model_slot=None):
    """Stage + run + finish for the reads `idx` of workload `w`.  Returns (results, ops_off, ops, timings)."""
    n = len(w["read_off"]) - 1
    whole = len(idx) == n and (n == 0 or (idx[0] == 0 and idx[-1] == n - 1))  # one rank: its shard is the set itself
    sub = w if whole else synth.take_reads(w, idx)
    t0 = time.perf_counter()
    b = ctx.stage_csr(params, sub["ref"], sub["ref_off"], sub["read"], sub["read_off"], sub["guide_ops"], sub["guide_off"],
                      model_slot=None if model_slot is None else np.ascontiguousarray(np.asarray(model_slot)[idx], dtype=np.int32),
                      ref_index=sub.get("ref_index"), guide_start=sub.get("guide_start"))
    t1 = time.perf_counter()
    try:
        kms = b.run()
        t2 = time.perf_counter()
        b.finish()
        t3 = time.perf_counter()
        res = b.results()
        off, ops = b.ops()
    finally:
        b.close()
    return res, off, ops, dict(stage_s=t1 - t0, run_s=t2 - t1, finish_s=t3 - t2, kernel_ms=kms)


def write_sam(path, w, nops, word_off, words, ref_names=None):
    """The realigned SAM: one record per read in input order, CIGAR = the realigner's ops (what
    realignSamFile3TargetFn splices in, utils.py:597-605), POS = where the guide's window starts on the reference.
    The cigars come packed (dist.index_packed_in_input_order) and are formatted natively."""
    from . import realign
    n = len(w["read_off"]) - 1
    cig, coff = realign.format_cigars_packed(word_off, nops, words)
    cig = cig.tobytes()
    read = np.ascontiguousarray(w["read"]).tobytes()
    ro = w["read_off"]
    ri = w.get("ref_index")
    gs = w.get("guide_start")
    n_refs = len(w["ref_off"]) - 1
    if ref_names is None:
        ref_names = ["ref_%d" % k for k in range(n_refs)]
    with open(path, "wb") as fh:
        fh.write(b"@HD\tVN:1.0\tSO:unsorted\n")
        for k in range(n_refs):
            fh.write(("@SQ\tSN:%s\tLN:%d\n" % (ref_names[k], int(w["ref_off"][k + 1] - w["ref_off"][k]))).encode())
        rn = [s.encode() for s in ref_names]
        lines = []
        for i in range(n):
            k = int(ri[i]) if ri is not None else i
            pos = (int(gs[i][0]) if gs is not None else 0) + 1
            lines.append(b"\t".join((b"read_%d" % i, b"0", rn[k], b"%d" % pos, b"255", cig[coff[i]:coff[i + 1]], b"*\t0\t0",
                                     read[ro[i]:ro[i + 1]], b"*")))
            if len(lines) >= 4096:
                fh.write(b"\n".join(lines) + b"\n")
                lines = []
        if lines:
            fh.write(b"\n".join(lines) + b"\n")


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


def run_job(ctx, params, w, out_dir=None, work=None, model_slot=None, device=None, group=None):
    """The whole job on this rank (collective: every rank of the process group calls it).  Without torch.distributed
    initialised it is the one-GPU job.  Returns on rank 0 a dict with the results in input order (status, score and the
    packed cigars n_ops / word_off / words -- dist.unpack_ops turns them into (op, length) pairs), the output paths and
    the stage timings of this rank; on other ranks the timings only."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if multi else 1
    rank = dist.get_rank(group) if multi else 0
    n = len(w["read_off"]) - 1
    if work is None:
        work = np.asarray(w["read_off"][1:]) - np.asarray(w["read_off"][:-1])
    mine = npd.shard_indices(work, world, rank)
    res, off, ops, tm = realign_shard(ctx, params, w, mine, model_slot)
    tm["cells"] = int(res["cells"].sum())
    t0 = time.perf_counter()
    if multi:
        got = npd.gather_to_root(npd.pack_results(mine, res["status"], res["score"], off, ops), device=device, group=group)
    else:
        got = [npd.pack_results(mine, res["status"], res["score"], off, ops)]
    tm["gather_s"] = time.perf_counter() - t0
    if rank != 0:
        return dict(timings=tm)
    t0 = time.perf_counter()
    status, score, nops, word_off, words = npd.index_packed_in_input_order(got, n)
    tm["merge_s"] = time.perf_counter() - t0
    out = dict(status=status, score=score, n_ops=nops, word_off=word_off, words=words, timings=tm)
    if out_dir is not None:
        t0 = time.perf_counter()
        os.makedirs(out_dir, exist_ok=True)
        out["sam"] = os.path.join(out_dir, "realigned.sam")
        out["xml"] = os.path.join(out_dir, "summary.xml")
        write_sam(out["sam"], w, nops, word_off, words)
        write_summary_xml(out["xml"], status, score, nops)
        tm["write_s"] = time.perf_counter() - t0
    return out
