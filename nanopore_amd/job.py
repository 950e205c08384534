"""The whole realignment job on one read set, sharded over the ranks of a node.

The reference's job is: one jobTree job per record of ONE SAM file (nanopore/analyses/utils.py:565-570), the temp cigar
files gathered in input order and spliced into a copy of the input SAM (utils.py:591-609).  Here: every rank takes a
contiguous range of the reads (`dist.shard_ranges`: balanced by read length), stages / realigns / closes it on its GPU
and writes the SAM records of its range at its own offset of the output file -- no data-path collective in either
direction --; one small gather (RCCL under backend nccl) brings the per-read scalars to rank 0 for the summary XML.  Used
by `bench.py --workload c3` (strong scaling, BASELINE.json configs[3]) and by the two-rank GPU test.
"""
import os
import time
import xml.etree.ElementTree as ET

import numpy as np

from . import dist as npd
from . import synth


def realign_shard(ctx, params, w, lo, hi, model_slot=None):
    """Stage + run + finish for the reads lo .. hi of workload `w`.  Returns (results, ops_off, packed cigar words, timings)."""
    n = len(w["read_off"]) - 1
    sub = w if (lo == 0 and hi == n) else synth.take_reads(w, np.arange(lo, hi))
    t0 = time.perf_counter()
    b = ctx.stage_csr(params, sub["ref"], sub["ref_off"], sub["read"], sub["read_off"], sub["guide_ops"], sub["guide_off"],
                      model_slot=None if model_slot is None else np.ascontiguousarray(np.asarray(model_slot)[lo:hi], dtype=np.int32),
                      ref_index=sub.get("ref_index"), guide_start=sub.get("guide_start"))
    t1 = time.perf_counter()
    try:
        kms = b.run()
        t2 = time.perf_counter()
        b.finish()
        t3 = time.perf_counter()
        res = b.results()
        off, words = b.ops_packed()
    finally:
        b.close()
    return res, off, words, dict(stage_s=t1 - t0, run_s=t2 - t1, finish_s=t3 - t2, kernel_ms=kms)


def sam_header(w, ref_names=None):
    n_refs = len(w["ref_off"]) - 1
    if ref_names is None:
        ref_names = ["ref_%d" % k for k in range(n_refs)]
    head = [b"@HD\tVN:1.0\tSO:unsorted\n"]
    for k in range(n_refs):
        head.append(("@SQ\tSN:%s\tLN:%d\n" % (ref_names[k], int(w["ref_off"][k + 1] - w["ref_off"][k]))).encode())
    return b"".join(head), ref_names


def sam_block(w, lo, hi, ops_off, words, ref_names):
    """The SAM records of reads lo .. hi as one bytes-like object: CIGAR = the realigner's ops (what realignSamFile3TargetFn
    splices in, utils.py:597-605), POS = where the guide's window starts on the reference.  The cigars come packed from the
    device and the records are formatted natively (npr_format_sam_records): at 50 k records per rank a Python loop over
    the records took longer than the DP."""
    from . import realign
    n = hi - lo
    nops = np.asarray(ops_off[1:]) - np.asarray(ops_off[:-1])
    ro = np.asarray(w["read_off"], dtype=np.int64)
    ri = w.get("ref_index")
    gs = w.get("guide_start")
    ref_index = np.asarray(ri[lo:hi], dtype=np.int32) if ri is not None else np.arange(lo, hi, dtype=np.int32)
    pos = (np.asarray(gs, dtype=np.int64)[lo:hi, 0] if gs is not None else np.zeros(n, dtype=np.int64)) + 1
    qnames = [b"read_%d" % i for i in range(lo, hi)]
    buf, _ = realign.format_sam_records(qnames, [s.encode() for s in ref_names], ref_index, pos, ops_off[:-1], nops, words,
                                        w["read"][ro[lo]:ro[hi]], ro[lo:hi + 1] - ro[lo])
    return memoryview(buf)


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
    """The whole job on this rank (collective: every rank of the process group calls it; without torch.distributed
    initialised it is the one-GPU job).

    Reads shard into CONTIGUOUS ranges balanced by work (dist.shard_ranges), so the output needs no data-path collective
    either: every rank formats the SAM records of its own range and writes them at its own offset of the one output file
    (an all_gather of the block sizes gives the offsets; the reference's single writer, utils.py:591-609, would serialise
    half a gigabyte of text behind eight GPUs).  Only the per-read scalars -- status, score, number of cigar operations --
    are gathered to rank 0, for the summary XML.  Returns on rank 0 a dict with those scalars in input order, the output
    paths and this rank's stage timings; on other ranks the timings only."""
    import torch
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if multi else 1
    rank = dist.get_rank(group) if multi else 0
    dev = torch.device("cpu") if device is None else torch.device(device)
    n = len(w["read_off"]) - 1
    if work is None:
        work = np.asarray(w["read_off"][1:]) - np.asarray(w["read_off"][:-1])
    bounds = npd.shard_ranges(work, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    res, off, words, tm = realign_shard(ctx, params, w, lo, hi, model_slot)
    tm["cells"] = int(res["cells"].sum())
    out = dict(timings=tm)
    if out_dir is not None:
        t0 = time.perf_counter()
        sam_path = os.path.join(out_dir, "realigned.sam")
        header, ref_names = sam_header(w)
        block = sam_block(w, lo, hi, off, words, ref_names)
        tm["format_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        sizes = [len(block)]
        if multi:
            t = torch.tensor([len(block)], dtype=torch.int64, device=dev)
            got = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(got, t, group=group)
            sizes = [int(g.item()) for g in got]
        if rank == 0:
            os.makedirs(out_dir, exist_ok=True)
            with open(sam_path, "wb") as fh:
                fh.write(header)
                fh.truncate(len(header) + sum(sizes))
        if multi:
            dist.barrier(group=group)
        fd = os.open(sam_path, os.O_WRONLY)
        try:
            os.pwrite(fd, block, len(header) + sum(sizes[:rank]))
        finally:
            os.close(fd)
        tm["write_s"] = time.perf_counter() - t0
        out["sam"] = sam_path
    # the one gather: per-read scalars for the summary
    t0 = time.perf_counter()
    mine = np.stack([res["status"].astype(np.float64), res["score"].astype(np.float64), (off[1:] - off[:-1]).astype(np.float64)], axis=1)
    if multi:
        got = npd.gather_to_root(mine.reshape(-1).view(np.uint8), device=device, group=group)
        if rank == 0:
            mine = np.concatenate([g.view(np.float64).reshape(-1, 3) for g in got])
        dist.barrier(group=group)  # every rank's block is on disk when rank 0 returns
    tm["gather_s"] = time.perf_counter() - t0
    if rank != 0:
        return dict(timings=tm)
    out["status"], out["score"], out["n_ops"] = mine[:, 0].astype(np.int64), mine[:, 1], mine[:, 2].astype(np.int64)
    if out_dir is not None:
        out["xml"] = os.path.join(out_dir, "summary.xml")
        write_summary_xml(out["xml"], out["status"], out["score"], out["n_ops"])
    return out
