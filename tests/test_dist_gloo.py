"""Multi-GPU path on CPU: world_size 2 -- and 8, the node BASELINE.json configs[3] names -- over gloo.  Reads shard with no data-path
collective; one variable-length gather brings packed results to rank 0 where the input order is restored."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nanopore_amd import dist as npd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_results(idx):
    """Deterministic per-read results a rank would produce for its reads."""
    status = (idx % 7 == 3).astype(np.int64) * -2
    score = idx / 100.0
    nops = idx % 4 + 1
    off = np.zeros(len(idx) + 1, dtype=np.int64)
    np.cumsum(nops, out=off[1:])
    ops = np.zeros((int(off[-1]), 2), dtype=np.int32)
    for k, i in enumerate(idx):
        for j in range(int(nops[k])):
            ops[off[k] + j] = (j % 3, int(i) + j + 1)
    return status, score, off, ops


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    work = np.arange(n_total)[::-1] % 13 + 1
    mine = npd.shard_indices(work, world, rank)
    status, score, off, ops = _fake_results(mine)
    got = npd.gather_to_root(npd.pack_results(mine, status, score, off, ops))
    if rank == 0:
        st, sc, ol = npd.merge_in_input_order(got, n_total)
        q.put((st.tolist(), sc.tolist(), [o.tolist() for o in ol]))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def _em_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nanopore_amd import em
    T = np.full((8, 25), float(rank + 1))
    E = np.arange(8 * 80, dtype=np.float64).reshape(8, 80) * (rank + 1)
    ll = np.array([-10.0 * (rank + 1)] + [0.0] * 7)
    T2, E2, ll2 = em.allReduceExpectations(T, E, ll)
    q.put((rank, T2.tolist(), float(E2[3, 7]), ll2.tolist()))
    dist.barrier()
    dist.destroy_process_group()


class _FakeCtx(object):
    def set_hmm(self, hmm, slot=0):
        self.hmm = hmm


class _FakeBatch(object):
    """Stands in for a staged GPU batch: expected counts and likelihood are a deterministic function of the installed
    model and of the rank's 'reads' (rank-dependent on purpose)."""

    def __init__(self, rank):
        self.ctx = _FakeCtx()
        self.rank = rank

    def expectations(self):
        t = np.asarray(self.ctx.hmm.transitions)
        e = np.asarray(self.ctx.hmm.emissions)
        T = np.zeros((8, 25))
        E = np.zeros((8, 80))
        T[0] = (t + 0.01) * (1 + self.rank) * np.arange(1, 26)
        E[0] = (e + 0.01) * (2 + self.rank)
        ll = np.zeros(8)
        ll[0] = -100.0 * (1 + self.rank) - float(np.sum(t * t)) * (3 - 2 * self.rank)
        return T, E, ll, 0.0


def _em_trials_worker(rank, world, port, tmp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nanopore_amd import em
    opt = em.Options()
    opt.trials, opt.iterations, opt.randomStart, opt.seed = 3, 2, True, None   # seed None: every rank draws its own start
    opt.outputXMLModelFile = None
    out = os.path.join(tmp, "model.txt")   # ONE path, as on a shared filesystem: rank 0 writes it, every rank returns once it exists
    best, trials, running = em.expectationMaximisationTrials(_FakeBatch(rank), out, opt)
    own = os.path.join(tmp, "own_%d.txt" % rank)
    best.write(own)
    assert open(out).read() == open(own).read()   # the file is the model THIS rank chose too
    q.put((rank, open(out).read(), [h.likelihood for h in trials], running))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_em_trials_choose_the_same_model_on_every_rank(tmp_path):
    """Every trial starts from rank 0's random model and every likelihood is summed over the ranks, so both ranks walk
    through identical models and write the same best one (nanopore_amd/em.py: broadcastModel, allReduceExpectations)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_em_trials_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] and got[0][2] == got[1][2] and got[0][3] == got[1][3]
    assert len(set(got[0][2])) == 3   # three different random starts, three different trials


@pytest.mark.timeout(120)
def test_em_expectations_all_reduce():
    """Sharded EM: expected counts and log-likelihoods are summed over ranks (the training loop's one collective)."""
    from nanopore_amd import em
    T, E, ll = np.ones((8, 25)), np.ones((8, 80)), np.zeros(8)
    a = em.allReduceExpectations(T, E, ll)            # no process group: identity
    assert a[0] is T and a[1] is E
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_em_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, T2, e37, ll2 in got:
        assert np.allclose(T2, 3.0) and e37 == (3 * 80 + 7) * 3 and ll2[0] == -30.0


def test_shards_partition_and_balance():
    work = np.random.default_rng(1).integers(1, 1000, size=1001)
    parts = [npd.shard_indices(work, 8, r) for r in range(8)]
    allidx = np.sort(np.concatenate(parts))
    assert (allidx == np.arange(1001)).all()
    loads = np.array([work[p].sum() for p in parts])
    assert loads.max() / loads.mean() < 1.05


def test_pack_unpack_round_trip():
    idx = np.array([5, 9, 2])
    status, score, off, ops = _fake_results(idx)
    rec, o = npd.unpack_results(npd.pack_results(idx, status, score, off, ops))
    assert rec["idx"].tolist() == [5, 9, 2] and (o == ops).all() and rec["nops"].tolist() == (off[1:] - off[:-1]).tolist()


@pytest.mark.timeout(120)
def test_two_rank_gather_restores_input_order():
    n_total = 37
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    st, sc, ol = q.get(timeout=100)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    status, score, off, ops = _fake_results(np.arange(n_total))
    assert st == status.tolist() and sc == score.tolist()
    for i in range(n_total):
        assert ol[i] == ops[off[i]:off[i + 1]].tolist()


@pytest.mark.timeout(300)
def test_eight_rank_gather_restores_input_order():
    """The same with the eight payloads of one node (BASELINE.json configs[3]): dist.shard_indices / gather_to_root / merge_in_input_order."""
    n_total = 203
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, n_total, q)) for r in range(8)]
    for p in procs:
        p.start()
    st, sc, ol = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    status, score, off, ops = _fake_results(np.arange(n_total))
    assert st == status.tolist() and sc == score.tolist()
    assert all(ol[i] == ops[off[i]:off[i + 1]].tolist() for i in range(n_total))


def _job_worker(rank, world, port, path, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["NPR_HOST_THREADS"] = str(max(1, 16 // world))  # what bench.py gives a rank of eight on a 16-core grant
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nanopore_amd import job, realign
    from test_pipeline_mock import FakeCtx, FakeSrc
    src = FakeSrc(n)
    src.header = b"@HD\tVN:1.0\n@SQ\tSN:ref\tLN:1000\n"
    work = np.random.default_rng(5).integers(200, 2000, size=n)      # every rank computes the same ranges from the same lengths
    bounds = npd.shard_ranges(work, world)
    out = job.run_source(src, realign.make_params(), bounds, path, ctxs=[FakeCtx() for _ in range(3)], gpu=0, chunk_bases=50000, coll_device="cpu")
    assert src.closed == len(src.staged) and out["timings"]["chunks"] >= 1
    if rank == 0:
        q.put((out["results"]["score"].tolist(), out["n_ops"].tolist(), bounds.tolist()))
    else:
        assert "results" not in out
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_eight_ranks_write_one_file_at_gathered_offsets(tmp_path):
    """job.run_source over eight ranks (stand-in batches, no GPU): every rank runs its contiguous range through the pipeline, the
    all_gather of the block sizes gives each its offset in the ONE output file, it writes its block there, and the per-read results
    arrive on rank 0 in input order -- the collectives of BASELINE.json configs[3] (nanopore/analyses/utils.py:591-609 gathers the
    per-read cigar files serially instead) with the world size of the node."""
    n, path = 4000, str(tmp_path / "out.bin")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, 8, port, path, n, q)) for r in range(8)]
    for p in procs:
        p.start()
    score, nops, bounds = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(bounds) == 9 and bounds[0] == 0 and bounds[-1] == n and all(b > a for a, b in zip(bounds[:-1], bounds[1:]))
    assert score == list(range(n)) and nops == [1] * n
    data = open(path, "rb").read()
    header = b"@HD\tVN:1.0\n@SQ\tSN:ref\tLN:1000\n"
    assert data[:len(header)] == header
    assert np.frombuffer(data[len(header):], dtype=np.int64).tolist() == list(range(n))  # every rank's block where it belongs
