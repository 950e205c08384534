"""Read sharding and result gathering for multi-GPU runs (one process per GPU, torch.distributed).

The path shards embarrassingly over reads -- the reference's own decomposition is one jobTree job per SAM
record (nanopore/analyses/utils.py:565-570) with the shared filesystem as the only "collective"
(rescoredCigar_%i.cig files gathered by realignSamFile3TargetFn, utils.py:591-609).  Here every rank
realigns its own shard with NO data-path collective; one variable-length gather to rank 0 (RCCL over xGMI
when the backend is nccl) brings the packed cigars / scores back for the single SAM / summary-XML writer,
and the input order is restored there (utils.py:597 zips by order).
"""
import numpy as np


def shard_indices(work, world_size, rank):
    """Indices of the reads rank `rank` owns: sort by descending work (cells ~ length) and deal round-robin,
    which balances cells rather than read counts."""
    order = np.argsort(-np.asarray(work), kind="stable")
    return np.sort(order[rank::world_size])


def shard_ranges(work, world_size):
    """Contiguous shards balanced by work: bounds[r] .. bounds[r + 1] are rank r's reads (input order kept inside and
    across ranks, so every rank can write its own block of the output file).  The prefix sum of the work is cut at
    multiples of total / world_size."""
    work = np.asarray(work, dtype=np.float64)
    n = len(work)
    cum = np.concatenate([[0.0], np.cumsum(work)])
    cuts = cum[-1] * np.arange(1, world_size) / world_size
    inner = np.searchsorted(cum, cuts, side="left")
    return np.concatenate([[0], np.minimum(inner, n), [n]]).astype(np.int64)


_REC = np.dtype([("idx", np.int64), ("status", np.int64), ("score", np.float64), ("nops", np.int64)])


def pack_results(indices, status, score, ops_off, ops):
    """Serialise one rank's results: header + per-read (global index, status, score, n_ops) + one 32-bit word per
    cigar op (length << 2 | op, the form the device stage hands the ops over in)."""
    n = len(indices)
    ops = np.ascontiguousarray(ops, dtype=np.int32).reshape(-1, 2)
    head = np.array([n, int(ops_off[-1])], dtype=np.int64)
    rec = np.zeros(n, dtype=_REC)
    rec["idx"] = indices
    rec["status"] = status
    rec["score"] = score
    rec["nops"] = np.asarray(ops_off[1:]) - np.asarray(ops_off[:-1])
    words = (ops[:, 1].astype(np.uint32) << np.uint32(2)) | ops[:, 0].astype(np.uint32)
    return np.concatenate([head.view(np.uint8), rec.view(np.uint8).reshape(-1), words.view(np.uint8)])


def unpack_results(buf):
    """-> (records, ops[(op, length)])"""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    n, nops = (int(v) for v in buf[:16].view(np.int64))
    rec = buf[16:16 + n * _REC.itemsize].view(_REC)
    words = buf[16 + n * _REC.itemsize:16 + n * _REC.itemsize + nops * 4].view(np.uint32)
    ops = np.empty((nops, 2), dtype=np.int32)
    ops[:, 0] = words & np.uint32(3)
    ops[:, 1] = words >> np.uint32(2)
    return rec, ops


def gather_to_root(payload, device=None, group=None, chunk_bytes=64 << 20):
    """Variable-length gather of one uint8 payload per rank to rank 0.

    all_gather of the byte counts (world x int64), then padded gathers of at most `chunk_bytes` per rank at a time (the
    cigars of 50 k reads are hundreds of megabytes: one padded gather of the whole payload would need world x max
    bytes of device memory on rank 0).  Returns the list of payloads on rank 0 and None elsewhere.  Works on any
    backend: tensors live on `device` (a CUDA device for nccl/RCCL, CPU for gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = torch.device("cpu") if device is None else torch.device(device)
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    size = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    sizes = [int(s.item()) for s in sizes]
    out = [np.empty(sz, dtype=np.uint8) for sz in sizes] if rank == 0 else None
    top = max(max(sizes), 1)
    for lo in range(0, top, chunk_bytes):
        cap = min(chunk_bytes, top - lo)
        send = torch.zeros(cap, dtype=torch.uint8, device=dev)
        part = payload[lo:lo + cap]
        if len(part):
            send[:len(part)] = torch.from_numpy(part).to(dev)
        recv = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
        dist.gather(send, recv, dst=0, group=group)
        if rank == 0:
            for r in range(world):
                k = max(0, min(cap, sizes[r] - lo))
                if k:
                    out[r][lo:lo + k] = recv[r][:k].cpu().numpy()
    return out


def index_packed_in_input_order(payloads, n_total):
    """Rank 0: the gathered payloads indexed in input (SAM) order WITHOUT moving the cigars: (status[n], score[n], n_ops[n],
    word_off[n], words) where read i's packed cigar (one uint32 per op, length << 2 | op) is
    words[word_off[i] .. word_off[i] + n_ops[i]).  Work is per read, not per cigar operation."""
    status = np.zeros(n_total, dtype=np.int64)
    score = np.zeros(n_total, dtype=np.float64)
    nops = np.zeros(n_total, dtype=np.int64)
    word_off = np.zeros(n_total, dtype=np.int64)
    seen = np.zeros(n_total, dtype=bool)
    parts, base = [], 0
    for p in payloads:
        buf = np.ascontiguousarray(p, dtype=np.uint8)
        n, total = (int(v) for v in buf[:16].view(np.int64))
        rec = buf[16:16 + n * _REC.itemsize].view(_REC)
        words = buf[16 + n * _REC.itemsize:16 + n * _REC.itemsize + total * 4].view(np.uint32)
        idx = rec["idx"]
        if seen[idx].any():
            raise ValueError("a read was realigned by two ranks")
        seen[idx] = True
        status[idx], score[idx], nops[idx] = rec["status"], rec["score"], rec["nops"]
        start = np.zeros(n, dtype=np.int64)
        np.cumsum(rec["nops"][:-1], out=start[1:])
        word_off[idx] = base + start
        parts.append(words)
        base += total
    if not seen.all():
        raise ValueError("%d reads came back from no rank" % int((~seen).sum()))
    words = parts[0] if len(parts) == 1 else (np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint32))
    return status, score, nops, word_off, words


def unpack_ops(nops, word_off, words):
    """(ops_off[n+1], ops[(op, length)]) in input order from the packed form (tests, small jobs)."""
    off = np.zeros(len(nops) + 1, dtype=np.int64)
    np.cumsum(nops, out=off[1:])
    src = np.repeat(np.asarray(word_off) - off[:-1], nops) + np.arange(int(off[-1]))
    w = np.asarray(words)[src]
    ops = np.empty((len(w), 2), dtype=np.int32)
    ops[:, 0] = w & np.uint32(3)
    ops[:, 1] = w >> np.uint32(2)
    return off, ops


def merge_csr_in_input_order(payloads, n_total):
    """Rank 0: every rank's payload unpacked into ONE result set in input (SAM) order (utils.py:597 zips by order):
    (status[n], score[n], ops_off[n+1], ops[(op, length)]).  Vectorised: no Python loop over reads."""
    recs, opss = zip(*[unpack_results(p) for p in payloads]) if payloads else ((), ())
    status = np.zeros(n_total, dtype=np.int64)
    score = np.zeros(n_total, dtype=np.float64)
    nops = np.zeros(n_total, dtype=np.int64)
    seen = np.zeros(n_total, dtype=bool)
    for rec in recs:
        idx = rec["idx"]
        if seen[idx].any():
            raise ValueError("a read was realigned by two ranks")
        seen[idx] = True
        status[idx], score[idx], nops[idx] = rec["status"], rec["score"], rec["nops"]
    if not seen.all():
        raise ValueError("%d reads came back from no rank" % int((~seen).sum()))
    off = np.zeros(n_total + 1, dtype=np.int64)
    np.cumsum(nops, out=off[1:])
    ops = np.empty((int(off[-1]), 2), dtype=np.int32)
    for rec, o in zip(recs, opss):
        k = rec["nops"]
        src = np.zeros(len(k) + 1, dtype=np.int64)
        np.cumsum(k, out=src[1:])
        dest = np.repeat(off[rec["idx"]] - src[:-1], k) + np.arange(int(src[-1]))
        ops[dest] = o
    return status, score, off, ops


def merge_in_input_order(payloads, n_total):
    """Rank 0: (status, score, list of per-read op arrays) in input order."""
    status, score, off, ops = merge_csr_in_input_order(payloads, n_total)
    return status, score, [ops[off[i]:off[i + 1]] for i in range(n_total)]
