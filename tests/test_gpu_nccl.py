"""The collectives of the sharded job and of sharded EM under backend `nccl` (= RCCL) with their tensors on the GPU.  The
box has one GPU and RCCL refuses two ranks on one device, so the process group has world size 1: every device-tensor branch
(`dist.gather_to_root`, the all_gather of the block sizes and the result gather of `job.run_source`,
`em.allReduceExpectations`, `em.broadcastModel`) runs through RCCL's own code, and the outputs must be those of the same calls
without torch.distributed.  Also: a plain `python bench.py --gpus 2` starts its own ranks (the driver's launch form).
Reference: nanopore/analyses/utils.py:565-570, :591-609 (what the shards and the gather replace), :471-531 (EM)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import MODEL_DIR, load_model_arrays

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HMM0 = os.path.join(MODEL_DIR, "blasr_hmm_0.txt")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _workload(n_reads):
    from nanopore_amd import synth
    T, E, _ = load_model_arrays()
    return synth.config_c3_shared(T, E, n_reads=n_reads, genome_len=300000)


def _em_batch(ctx):
    from nanopore_amd import realign as R, synth
    T, E, _ = load_model_arrays()
    w = synth.make_workload(77, 48, 600, T, E, flank=0)
    P = R.make_params(band_mode=R.BAND_ANCHOR, split_threshold=300, mode=R.MODE_EXPECTATIONS)
    return ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])


def _release_device_memory(gpu_ctx):
    """The processes these tests start need the HBM this session's earlier (full-size) tests left in the device's scratch and in the
    contexts' caches (npr_ctx_option NPR_OPT_RELEASE_SCRATCH)."""
    from nanopore_amd import job
    for c in [gpu_ctx] + [c for pool in job._ctx_pool.values() for c in pool]:
        if getattr(c, "_h", None):
            c.release_scratch()


def _nccl_rank(rank, port, n_reads, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from nanopore_amd import dist as npd, em, job, realign as R, synth
    from nanopore_amd.hmm import Hmm
    assert str(job._collective_device(dist, None, 0)) == "cuda:0" and str(em._collectiveDevice(None, None)).startswith("cuda")
    # 1. the variable-length gather on device tensors, in several chunks
    rng = np.random.default_rng(3)
    payload = rng.integers(0, 256, 1_000_003, dtype=np.uint8)
    got = npd.gather_to_root(payload, device="cuda:0", chunk_bytes=1 << 18)
    assert len(got) == 1 and np.array_equal(got[0], payload)
    # 2. the job from arrays and from files: collectives default to the GPU under nccl
    w, W = _workload(n_reads)
    ctx = R.Context(0)
    ctx.set_hmm(Hmm.loadHmm(HMM0))
    P = R.make_params(band_mode=R.BAND_FIXED, fixed_width=W)
    out = job.run_job(ctx, P, w, out_dir=os.path.join(out_dir, "arrays"))
    np.savez(os.path.join(out_dir, "arrays.npz"), status=out["status"], score=out["score"], n_ops=out["n_ops"])
    sam, fa = os.path.join(out_dir, "in.sam"), os.path.join(out_dir, "ref.fa")
    synth.write_workload_files(w, sam, fa)
    res = job.realign_sam_file(sam, os.path.join(out_dir, "files.sam"), fa, hmm=HMM0, params=P, gpu=0)
    assert res["records"] == n_reads and (res["results"]["status"] == 0).all() and res["timings"]["gather_s"] > 0
    # 3. sharded EM: the all-reduce of the expectations and the model broadcast on the GPU
    h = Hmm.loadHmm(HMM0)
    b = _em_batch(ctx)
    T, E, ll, _ = b.expectations()
    T2, E2, ll2 = em.allReduceExpectations(T, E, ll)
    assert np.array_equal(T, T2) and np.array_equal(E, E2) and np.array_equal(ll, ll2) and T2.shape == np.asarray(T).shape
    h2 = em.broadcastModel(h.copy())
    assert h2.transitions == h.transitions and h2.emissions == h.emissions
    opt = em.Options()
    opt.trials, opt.iterations, opt.seed, opt.outputTrialHmms = 2, 3, 11, False
    opt.outputXMLModelFile = os.path.join(out_dir, "hmm.xml")
    best, trials, running = em.expectationMaximisationTrials(b, os.path.join(out_dir, "hmm.txt"), opt)
    np.save(os.path.join(out_dir, "running.npy"), np.array(running))
    b.close()
    dist.barrier()
    ctx.close()
    job.close_contexts()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_job_and_em_collectives_run_on_device_tensors_under_nccl(gpu_ctx, tmp_path):
    import torch.multiprocessing as mp
    from nanopore_amd import em, job, realign as R, synth
    from nanopore_amd.hmm import Hmm
    n_reads = 640
    out_dir = str(tmp_path / "nccl")
    os.makedirs(out_dir)
    _release_device_memory(gpu_ctx)
    mp.spawn(_nccl_rank, args=(_free_port(), n_reads, out_dir), nprocs=1, join=True)
    # the same calls without torch.distributed
    w, W = _workload(n_reads)
    gpu_ctx.set_hmm(Hmm.loadHmm(HMM0))
    P = R.make_params(band_mode=R.BAND_FIXED, fixed_width=W)
    one = job.run_job(gpu_ctx, P, w, out_dir=str(tmp_path / "plain"))
    z = np.load(os.path.join(out_dir, "arrays.npz"))
    for k in ("status", "score", "n_ops"):
        assert np.array_equal(z[k], one[k]), k
    plain = open(one["sam"], "rb").read()
    assert open(os.path.join(out_dir, "arrays", "realigned.sam"), "rb").read() == plain
    assert open(os.path.join(out_dir, "arrays", "summary.xml"), "rb").read() == open(one["xml"], "rb").read()
    # files -> file under nccl: the records of the input with the same cigars (header differs: @SQ names of the written FASTA)
    recs = lambda t: [l.split(b"\t")[5] for l in t.split(b"\n") if l and not l.startswith(b"@")]
    assert recs(open(os.path.join(out_dir, "files.sam"), "rb").read()) == recs(plain)
    # EM: the same trials, the same running likelihoods
    b = _em_batch(gpu_ctx)
    opt = em.Options()
    opt.trials, opt.iterations, opt.seed, opt.outputTrialHmms = 2, 3, 11, False
    best, trials, running = em.expectationMaximisationTrials(b, str(tmp_path / "hmm.txt"), opt)
    b.close()
    got = np.load(os.path.join(out_dir, "running.npy"))
    assert got.shape == (2, 4) and np.allclose(got, np.array(running), rtol=1e-9)
    assert open(os.path.join(out_dir, "hmm.txt")).read() == open(str(tmp_path / "hmm.txt")).read()
    gpu_ctx.set_hmm(Hmm.loadHmm(HMM0))


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("ranks,reads", [(2, 1024), (8, 512)], ids=["two_ranks", "eight_ranks"])
def test_plain_bench_command_line_starts_its_own_ranks(gpu_ctx, tmp_path, ranks, reads):
    """`python bench.py --gpus N ...` without a launcher (the driver's form): N ranks are spawned, share cuda:0 through the
    test hook (gloo collectives), rank 0 prints ONE JSON line with the weak-scaling headline and the strong-scaling job beside
    it, and the exit code is 0.  N = 8: the rehearsal of the node BASELINE.json configs[3] names -- eight processes map one SAM,
    each with 16 // 8 host threads, eight offsets are gathered, eight blocks land in one file."""
    from nanopore_amd import job
    _release_device_memory(gpu_ctx)
    env = dict(os.environ, NPR_BENCH_SHARE_GPU="1", NPR_BENCH_ALSO_READS=str(max(1024, 8 * ranks)), TMPDIR=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NPR_HOST_THREADS"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--reads", str(reads)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1400)
    err = p.stderr.decode()
    assert p.returncode == 0, "\n".join([l for l in err.split("\n") if "[rank0]" in l][-25:]) + err[-1500:]
    lines = [l for l in p.stdout.decode().split("\n") if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["scaling"] == "weak" and out["ok_reads"] == reads and out["value"] > 0
    also = [a for a in out["also"] if a.get("scaling") == "strong"]
    assert len(also) == 1 and also[0]["n_gpus"] == ranks and also[0]["ok_reads"] == max(1024, 8 * ranks)
    assert also[0]["speedup_vs_n1"] > 0 and also[0]["n1_ms_per_step"] > 0
