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


def pack_results(indices, status, score, ops_off, ops):
    """Serialise one rank's results: header + per-read (global index, status, score, n_ops) + op pairs."""
    n = len(indices)
    head = np.array([n, int(ops_off[-1])], dtype=np.int64)
    rec = np.zeros(n, dtype=[("idx", np.int64), ("status", np.int64), ("score", np.float64), ("nops", np.int64)])
    rec["idx"] = indices
    rec["status"] = status
    rec["score"] = score
    rec["nops"] = ops_off[1:] - ops_off[:-1]
    return np.concatenate([head.view(np.uint8), rec.view(np.uint8).reshape(-1),
                           np.ascontiguousarray(ops, dtype=np.int32).reshape(-1).view(np.uint8)])


def unpack_results(buf):
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    n, nops = (int(v) for v in buf[:16].view(np.int64))
    rec_dt = np.dtype([("idx", np.int64), ("status", np.int64), ("score", np.float64), ("nops", np.int64)])
    rec = buf[16:16 + n * rec_dt.itemsize].view(rec_dt)
    ops = buf[16 + n * rec_dt.itemsize:16 + n * rec_dt.itemsize + nops * 8].view(np.int32).reshape(-1, 2)
    return rec, ops


def gather_to_root(payload, device=None, group=None):
    """Variable-length gather of one uint8 payload per rank to rank 0.

    all_gather of the byte counts (world x int64), then one padded gather.  Returns the list of payloads on
    rank 0 and None elsewhere.  Works on any backend: tensors live on `device` (a CUDA device for nccl/RCCL,
    CPU for gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = torch.device("cpu") if device is None else torch.device(device)
    size = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    send = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if len(payload):
        send[:len(payload)] = torch.from_numpy(np.ascontiguousarray(payload)).to(dev)
    recv = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0, group=group)
    if rank != 0:
        return None
    return [r[:s].cpu().numpy() for r, s in zip(recv, sizes)]


def merge_in_input_order(payloads, n_total):
    """Rank 0: unpack every rank's payload and restore the input (SAM) order."""
    status = np.zeros(n_total, dtype=np.int64)
    score = np.zeros(n_total, dtype=np.float64)
    ops = [None] * n_total
    for p in payloads:
        rec, o = unpack_results(p)
        pos = 0
        for r in rec:
            i = int(r["idx"])
            status[i] = r["status"]
            score[i] = r["score"]
            ops[i] = o[pos:pos + int(r["nops"])].copy()
            pos += int(r["nops"])
    return status, score, ops
