#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X realigner.

Metric (BASELINE.json): banded pair-HMM DP cells/s (whole job), plus reads/s.

    python bench.py --gpus N --steps K --warmup W [--workload northstar|c2|anchor|c3]

Resident workloads (northstar -- the default --, c2, anchor): a "step" is one pass of the hot path over one synthetic
batch whose inputs are already resident in HBM: npr_batch_run (forward + backward + posterior extraction of every read)
+ npr_batch_finish (MEA chain and cigar on the device, ops and per-read results to the host).  N > 1 is launched by
torch.distributed.run, one rank per GPU; every rank realigns its own batch of the named shape with no data-path
collective (weak scaling); one chunked RCCL gather of the packed results to rank 0 closes the job, outside the timed steps.
  northstar  ~10 kb reads x 50 kb reference slices, band 200, blasr_hmm_0: the shape the north-star target is quoted on
  c2         BASELINE.json configs[1]: 1 k reads x 1 kb, band 100
  anchor     ~8 kb reads in the reference's own band (nanopore/analyses/utils.py:587: anchors +- 10, trim 14, split 3000)

`--workload c3` is the STRONG-scaling job of BASELINE.json configs[2] / configs[3]: ONE set of 50 k reads (~8 kb, one
shared 4.6 Mb contig) sharded over the ranks as the reference shards one SAM file over jobTree jobs
(utils.py:565-570); a step is the whole job FILES -> FILE (nanopore_amd/job.py::realign_sam_file, what
analyses.utils.realignSamFile runs): every rank maps the SAM + FASTA, parses the records of its contiguous shard natively,
keeps two batches in flight (stage = band planning + upload, DP, finish = MEA chain + cigar) and writes its block of the
realigned SAM at its offset of the one output file; the per-read results are gathered to rank 0 (RCCL) inside the timed
region (utils.py:557-609).  `--resident-arrays` runs the same pipeline from arrays already in host memory (round 2's job).
value = cells of the whole set / that time; reads_per_s likewise.  The default run (any N) carries this job as an `also`
entry, so that a --gpus 1/2/4/8 series measures the strong-scaling target next to the weak-scaling headline.

`--workload em` is the Baum-Welch E-step (npr_batch_expectations: what the trainer runs 3 x 100 times per model,
utils.py:509-523) over one resident batch in the trainer's own band (anchors, splitMatrixBiggerThanThis 300, utils.py:511).  The default
run at N = 1 carries it as an `also` entry too (seven timed steps, after the reference's own band and the rescore mode).

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The ROCm runtime multiplexes a process's HIP streams over four hardware queues by default, and two streams on one queue run their
# kernels in submission order.  The files -> file job keeps three batches in flight on 21 streams: with four queues the MEA stage of
# one chunk and the DP pass of the next regularly share one (tools/queues_of.py), with eight they do not -- 403 -> 393 ms per job.
# Read by the runtime when it starts; a deployment setting (the job's entry points ask for it through _lib.want_hw_queues(), which
# leaves a host's own choice alone) -- set here because this script touches the GPU through torch before it calls the job.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E (guides/MI355X_MICROARCH.md); ~6300 GB/s achievable
DECLARED_BYTES_PER_CELL = 40.0  # all five fp32 states stored + reloaded.  This design keeps only the match
                                # state for the backward sweep, so its algorithmic traffic is the next two constants.
BYTES_PER_CELL = 16.0       # (mantissa, exponent) of the match state: 8 B stored by the forward sweep + 8 B reloaded
BYTES_PER_PAIR = 12.0       # one sparse posterior triple (x, y, p) written per pair >= 0.01
EM_BYTES_PER_CELL = 48.0    # E-step: all five forward states kept for the backward sweep (8 B match pair + 16 B of the
                            # other four), stored once and reloaded once
SIMDS = 256 * 4             # 256 CUs x 4 SIMDs
NOMINAL_CYCLES_PER_VALU = 2.0  # one wave64 VALU instruction per 2 cycles per SIMD: the 157 TF fp32 vector peak

# kernel class (npr_batch_class_stats) -> kernel name in profiles/kernel_table.json
CLASS_KERNEL = {0: "k_dp_stair<1>", 1: "k_dp_stair<2>", 2: "k_dp_stair<4>", 3: "k_dp_wide", 4: "k_dp_wide", 5: "k_dp_wide",
                6: "k_dp_wide", 7: "k_dp_generic", 8: "k_dp_generic", 9: "k_dp_generic", 10: "k_dp_generic", 11: "k_dp_tile<2>",
                12: "k_dp_mid_rs<1>", 13: "k_dp_mid_rs<2>", 14: "k_dp_mid_rs<4>", 15: "k_dp_rs<1>", 16: "k_dp_rs<2>", 17: "k_dp_rs<4>", 18: "k_dp_tile_cs"}
RS_BYTES_PER_CELL = 8.0     # k_dp_rs / k_dp_mid_rs (row-scaled arithmetic: the exponent is per row, not per cell): 4 B stored + 4 B reloaded
                            # (k_dp_mid_rs: the forward sweep stores the rows up to the cut and the backward sweep the rows above it: the same bytes)
EM_CLASS_KERNEL = {0: "k_em_stair<1>", 1: "k_em_stair<2>", 2: "k_em_stair<4>", 11: "k_em_tile<2>", 18: "k_dp_tile_cs<EM>"}
EM_CS_BYTES_PER_CELL = 32.0  # k_dp_tile_cs's E-step instance (round 6): 4 B match value + 12 B for the other four states as 24-bit floats, stored once
                             # and reloaded once (the exponents are per lane and block of sixteen rows)


def load_model(name="blasr_hmm_0.txt"):
    from nanopore_amd.hmm import Hmm
    return Hmm.loadHmm(os.path.join(ROOT, "nanopore_amd", "mappers", name))


def kernel_table():
    """Per-kernel counters measured with rocprofv3 --pmc (tools/pmc_passes.sh), committed under profiles/: VALU
    instructions per cell, issue cycles per VALU instruction, HBM bytes per cell.  The roofline block is derived from
    these and the launch time measured live."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "kernel_table.json")))
    except (OSError, ValueError):
        return {}


def build_workload(name, n_reads, rank):
    from nanopore_amd import synth
    h = load_model()
    if name == "northstar":
        w, W = synth.config_north_star(h.transitions, h.emissions, n_reads=n_reads, seed=1003 + 7919 * rank)
        label = ("synthetic ~10kb reads x 50kb reference slices, band 200, blasr_hmm_0 (north-star shape); guides carry "
                 "their window's coordinates inside the slice, as the exonerate cigar fed to cactus_realign does")
    elif name == "c2":
        w, W = synth.make_workload(1001 + 7919 * rank, n_reads, 1000, h.transitions, h.emissions), 100
        label = "synthetic 1k reads x 1kb, band 100, blasr_hmm_0 (BASELINE.json configs[1])"
    elif name == "anchor":
        w, W = synth.make_workload(1004 + 7919 * rank, n_reads, 8000, h.transitions, h.emissions), 0
        label = ("synthetic ~8kb reads, the reference's own band: anchors +- diagonalExpansion 10, 14 trimmed columns, "
                 "splitMatrixBiggerThanThis 3000 (nanopore/analyses/utils.py:587), blasr_hmm_0")
    elif name == "rescore":
        # (jitter 0: the guide is the alignment being scored, as the analysis scores the mapper's or the realigner's own; with the indels slid by
        # +- 20 columns a guide can leave the band of +- 10 around its own anchors)
        w, W = synth.make_workload(1006 + 7919 * rank, n_reads, 8000, h.transitions, h.emissions, jitter=0), -1
        label = ("synthetic ~8kb reads, NPR_MODE_RESCORE_ORIGINAL with the analyses' call parameters: --rescoreOriginalAlignment "
                 "--diagonalExpansion=10 --splitMatrixBiggerThanThis=100, blasr_hmm_0 (nanopore/analyses/alignmentUncertainty.py:41: the analysis "
                 "every experiment runs by default, pipeline.py:81); score = mean posterior over the guide's M columns")
    else:
        raise SystemExit("unknown workload %s" % name)
    return h, w, W, label


def param_kwargs(W, mod):
    """npr_params of a workload as keyword arguments, for the library (mod = nanopore_amd.realign) and for the oracle (mod = oracle.oracle).
    W > 0: fixed band; 0: the reference's own realign call (anchors +- 10, trim 14, split 3000; utils.py:587); -1: the analyses' rescore call."""
    if W > 0:
        return dict(band_mode=mod.BAND_FIXED, fixed_width=W)
    if W == 0:
        return dict(band_mode=mod.BAND_ANCHOR)
    return dict(band_mode=mod.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=100, mode=mod.MODE_RESCORE_ORIGINAL)


def make_params(W):
    from nanopore_amd import realign as R
    kw = param_kwargs(W, R)
    if W <= 0:
        kw["max_pairs_per_base"] = 24  # (the anchor bands' wide rectangles hold more pairs per base than the default list has room for)
    return R.make_params(**kw)


def usable_cpus():
    """CPUs this process may use: os.cpu_count() capped by the scheduler affinity and the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(period) > 0:
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(h, w, W, cells_per_read, budget_s=9.0):
    """The build's fp64 log-space CPU oracle (kind "port": the reference binary cactus_realign is absent from
    the snapshot) timed on this host's cores over bounded samples of the same workload, in the three
    configurations SURVEY.md 8(d) names: one thread, the reference's 4 workers (/root/reference/Makefile:1: jobTree
    --maxThreads=4), and every core the cgroup grants.  `value` is the all-cores figure."""
    from oracle import oracle as orc
    from nanopore_amd.realign import encode
    cores = usable_cpus()
    oh = orc.make_hmm(h.transitions, h.emissions)
    # W = 0: the reference's own call parameters (anchors +- 10, trim 14, splitMatrixBiggerThanThis 3000; utils.py:587); -1: the analyses' rescore call
    P = orc.make_params(**param_kwargs(W, orc))

    def run(k, threads):
        if w.get("guide_start") is not None:
            # the oracle takes the guide's window of every slice (what npr_batch_create_at does on its side)
            lead, ilen = w["guide_start"][:k, 0], w["interval_len"][:k]
            X = encode(b"".join(bytes(w["ref"][w["ref_off"][i] + lead[i]:w["ref_off"][i] + lead[i] + ilen[i]]) for i in range(k)))
            x_off = np.concatenate([[0], np.cumsum(ilen)]).astype(np.int64)
        else:
            X = encode(bytes(w["ref"][:w["ref_off"][k]]))
            x_off = w["ref_off"][:k + 1]
        Y = encode(bytes(w["read"][:w["read_off"][k]]))
        t0 = time.time()
        r = orc.realign_batch(oh, P, X, x_off, Y, w["read_off"][:k + 1],
                              w["guide_ops"][:w["guide_off"][k]], w["guide_off"][:k + 1], precision=0,
                              threads=threads, native=True)
        return r, time.time() - t0

    # pilot (one read per core) to size the timed samples for ~budget_s of wall time each
    n = len(cells_per_read)
    mean_cells = max(float(np.mean(cells_per_read)), 1.0)
    k0 = min(n, cores)
    r, dt = run(k0, cores)
    rate_all = float(r["cells"].sum()) / max(dt, 1e-3)
    variants, keep = [], None
    for threads in sorted({1, min(4, cores), cores}):
        guess = rate_all * threads / cores
        k = int(min(n, max(threads, round(guess * budget_s / mean_cells))))
        r, dt = run(k, threads)
        cells = int(r["cells"].sum())
        variants.append({"threads": threads, "value": cells / dt, "unit": "cells/s", "cells_per_s_per_core": cells / dt / threads,
                         "reads": k, "cells": cells, "seconds": dt})
        if threads == cores:
            keep = (r, k, cells, dt)
    r, k, cells, dt = keep
    return {"value": cells / dt, "unit": "cells/s", "cores": cores, "kind": "port", "cpu": cpu_model(), "variants": variants,
            "sample": "first %d reads of the same batch (%d cells), fp64 log-space oracle (this build's own restatement of "
                      "cactus_realign: the reference binary is absent, parity unpinned), OpenMP over reads on %d threads "
                      "(os.cpu_count() %d, capped by the cgroup CPU quota) of a %s, gcc -O3 -march=native, %.1f s; `variants`: "
                      "the same on 1 thread, on the reference's 4 workers (/root/reference/Makefile:1) and on all granted cores"
                      % (k, cells, cores, os.cpu_count() or 1, cpu_model(), dt)}, r


def roofline_block(cells, pairs, kms, class_cells, clock_hz):
    """The dominant DP kernel against its ceilings.  HBM: bytes this design has to move (16 B/cell + 12 B/pair) over the
    launch time measured live (HIP events on the library's stream) against the 8 TB/s peak.  The binding resource is
    VALU issue: instructions per cell and issue cycles per instruction come from the committed PMC table of that kernel."""
    dom = int(np.argmax(class_cells)) if len(class_cells) else 1
    kname = CLASS_KERNEL.get(dom, "k_dp")
    tab = kernel_table().get(kname, {})
    t = kms * 1e-3
    rs_kernel = kname.startswith("k_dp_rs") or kname.startswith("k_dp_mid_rs") or kname == "k_dp_tile_cs"
    per_cell = RS_BYTES_PER_CELL if rs_kernel else BYTES_PER_CELL
    achieved = (per_cell * cells + BYTES_PER_PAIR * pairs) / t / 1e9
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
           "traffic": None, "kernel": kname, "kernel_ms": kms,
           "kernel_share_of_cells": float(class_cells[dom]) / max(float(np.sum(class_cells)), 1.0) if len(class_cells) else 1.0,
           "algorithmic_bytes": ("%g B/cell (match state: %g B stored by one sweep + %g B reloaded by the other%s%s) + "
                                 "%g B/posterior pair (%d pairs); DESIGN.md section 4"
                                 % (per_cell, per_cell / 2, per_cell / 2,
                                    "; k_dp_mid_rs runs a read's sweeps on two wavefronts that meet in the middle: the forward one stores the rows up to the cut, "
                                    "the backward one the rows above it, each reloads the other's -- the bytes of k_dp_rs" if kname.startswith("k_dp_mid") else "",
                                    ("; one exponent per anti-diagonal row instead of one per cell: half of round 2's 16 B/cell, so the same "
                                     "cells/s is half the HBM fraction -- see round2_16B_equiv" if per_cell < BYTES_PER_CELL else ""),
                                    BYTES_PER_PAIR, pairs)),
           "binding": "vector and scalar issue, not HBM (see valu)",
           "round2_16B_equiv": {"GBps": (BYTES_PER_CELL * cells + BYTES_PER_PAIR * pairs) / t / 1e9,
                                "frac": (BYTES_PER_CELL * cells + BYTES_PER_PAIR * pairs) / t / 1e9 / HBM_PEAK_GBPS,
                                "note": "the same launch priced at the 16 B/cell of the per-cell-exponent kernels (k_dp_stair, rounds 1-2): "
                                        "what `frac` would read had the bytes not been halved; comparison only"},
           "declared_40B_equiv": {"GBps": DECLARED_BYTES_PER_CELL * cells / t / 1e9,
                                  "note": "SURVEY.md 8d prices a cell at 40 B (all five states stored and reloaded); this design "
                                          "stores only the match state, so that figure is NOT traffic it generates -- kept for "
                                          "comparison only, not a roofline fraction"}}
    if tab.get("hbm_bytes_per_cell") is not None:
        out["traffic"] = tab["hbm_bytes_per_cell"] * cells / 1e9
        out["traffic_unit"] = "GB per launch = PMC bytes per cell of this kernel (%s) x cells of this launch" % tab.get("hbm_source", "profiles/")
        out["traffic_frac"] = out["traffic"] / t / HBM_PEAK_GBPS
    if tab.get("valu_insts_per_cell") is not None:
        insts = tab["valu_insts_per_cell"] * cells
        clock = tab.get("clock_hz_under_load", clock_hz)  # the clock the chip holds under this kernel (PMC run), not the 2.4 GHz peak
        simd_cycles = t * clock * SIMDS
        counter_ratio = insts * tab["cycles_per_valu_inst"] / simd_cycles
        out["valu"] = {"insts_per_cell": tab["valu_insts_per_cell"], "issue_cycles_per_inst": tab["cycles_per_valu_inst"],
                       "clock_hz": clock, "peak_clock_hz": clock_hz, "simd_cycles": simd_cycles,
                       "busy_frac": min(1.0, counter_ratio), "counter_ratio": counter_ratio,
                       "nominal_frac": insts * NOMINAL_CYCLES_PER_VALU / simd_cycles,
                       "useful_fp32_frac_of_peak": insts * NOMINAL_CYCLES_PER_VALU / (t * clock_hz * SIMDS),
                       "simd_cycles_per_valu_inst": simd_cycles / insts,
                       "salu_insts_per_cell": tab.get("salu_insts_per_cell"),
                       "note": "counter_ratio = VALU instructions x issue cycles per instruction (4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU: "
                               "the counter ticks in quad-cycles and over-counts instructions that issue in fewer) / SIMD-cycles of the "
                               "launch at the clock held under load; busy_frac caps it at 1 (the pipes are saturated when it reaches 1); "
                               "nominal_frac prices every instruction at the 2-cycle wave64 issue peak instead: the distance to the "
                               "157 TF fp32 vector peak that a cheaper instruction mix could still close; "
                               "simd_cycles_per_valu_inst = SIMD-cycles of this launch per VALU instruction (what an instruction "
                               "costs all told); salu_insts_per_cell: scalar instructions, which share the CU's one scalar unit and "
                               "the SIMDs' issue slots (tools/issue_mix.hip)",
                       "source": tab.get("valu_source", "profiles/")}
    return out


def gpu_clock_hz():
    import torch
    try:
        khz = torch.cuda.get_device_properties(torch.cuda.current_device()).clock_rate
        if khz and khz > 1e5:
            return float(khz) * 1e3
    except Exception:
        pass
    return 2.4e9


ALSO_STEPS = int(os.environ.get("NPR_BENCH_ALSO_STEPS") or 7)  # timed steps of every secondary line (two or three decide nothing inside a 5 % box spread)


def spread(ms):
    """median and min - max of a list of per-step times: what a secondary line reports next to its mean"""
    v = np.asarray(ms, dtype=np.float64)
    return {"n": int(v.size), "median_ms": float(np.median(v)), "min_ms": float(v.min()), "max_ms": float(v.max())}


def timed_resident(ctx, h, w, W, steps, warmup, sync, options=None):
    """Stages one batch and times `steps` passes of run + finish.  Returns everything the line needs."""
    params = make_params(W)
    for k, v in (options or {}).items():  # (context options that pick a kernel take effect when a batch is staged)
        ctx.set_option(k, v)
    t0 = time.perf_counter()
    batch = ctx.stage_csr(params, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"],
                          guide_start=w.get("guide_start"))
    first_create_s = time.perf_counter() - t0
    st = batch.stats()
    for _ in range(warmup):
        batch.run()
        batch.finish()
    sync()
    t0 = time.perf_counter()
    kernel_ms, finish_ms, step_ms = [], [], []
    for _ in range(steps):
        ts = time.perf_counter()
        kernel_ms.append(batch.run())  # blocks until the DP launch has finished on the library's stream
        tf = time.perf_counter()
        batch.finish()
        te = time.perf_counter()
        finish_ms.append((te - tf) * 1e3)
        step_ms.append((te - ts) * 1e3)
    sync()
    elapsed = time.perf_counter() - t0
    res = batch.results()
    _, class_cells = batch.class_stats()
    return dict(batch=batch, params=params, stats=st, elapsed=elapsed, kernel_ms=float(np.mean(kernel_ms)), kernel_ms_all=kernel_ms,
                finish_ms=float(np.mean(finish_ms)), step_ms=step_ms, res=res, class_cells=np.asarray(class_cells), first_create_s=first_create_s)


def from_cold(ctx, w, params, cells):
    """One more batch of the same inputs from host buffers, as a pipeline that stages batch after batch would see it: stage
    (plan points on the host, H2D, band rows / schedules on the device) + run + finish, with the context's scratch arena
    and its cache of released device buffers warm (one untimed stage-and-release first: hipMalloc of the gigabyte-sized
    arrays costs tens of milliseconds).  PCIe-inclusive: reported next to the line, never as `value`."""
    ctx.stage_csr(params, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"],
                  guide_start=w.get("guide_start")).close()
    t0 = time.perf_counter()
    b = ctx.stage_csr(params, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"],
                      guide_start=w.get("guide_start"))
    t1 = time.perf_counter()
    b.run()
    t2 = time.perf_counter()
    b.finish()
    t3 = time.perf_counter()
    b.close()
    return {"value": cells / (t3 - t0), "unit": "cells/s", "create_s": t1 - t0, "run_s": t2 - t1, "finish_s": t3 - t2,
            "note": "host buffers -> results: npr_batch_create (plan, pack, H2D) + npr_batch_run + npr_batch_finish"}


def self_launch(n):
    """Re-executes this command line under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1, a free
    port) and returns its exit code; rank 0's JSON line goes to this process's stdout unchanged."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="northstar", choices=["northstar", "c2", "c3", "anchor", "em", "rescore"])
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (resident workloads; default 24576 northstar = four per resident "
                    "wavefront slot (rounds 1-2: 12288; the launch lasts as long as its longest chain of reads, which two per slot leave "
                    "10 %% above the mean), 1000 c2, 8192 anchor, 6144 em) or in the whole set (c3; default 50000)")
    ap.add_argument("--resident-arrays", action="store_true", help="c3: the job from arrays in host memory instead of from files")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary lines of the default run (the reference's own band; the "
                    "files -> file strong-scaling job of configs[2]/[3])")
    ap.add_argument("--ab", default="", help="NAME=a,b: a paired A/B inside ONE process on ONE box -- the steps alternate between the two values of a context "
                    "option (nanopore_amd._lib.OPTIONS, e.g. tile_rs=0,2; resident workloads: one staged batch per value) or of the job's NPR_JOB_OVERLAP "
                    "(job_overlap=1,2 with --workload c3); prints per-value median / min - max and the paired differences with their spread")
    ap.add_argument("--ab-pairs", type=int, default=7)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # a plain `python bench.py --gpus N ...`: start the N ranks ourselves (one process per GPU, the launch line the
        # driver uses) and hand their one JSON line through
        raise SystemExit(self_launch(args.gpus))
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d: launch with python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ... (or plainly: python bench.py --gpus %d)"
                         % (world, args.gpus, args.gpus, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # test hook: NPR_BENCH_SHARE_GPU=1 lets N ranks share cuda:0 with gloo collectives, to exercise the
    # multi-rank code path on a one-GPU box (RCCL refuses two ranks on one device); never used by the driver
    share = os.environ.get("NPR_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll_dev = "cpu" if share else "cuda:%d" % local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if world > 1:  # the ranks of one node share its host cores: split them instead of oversubscribing each rank's pool
        os.environ.setdefault("NPR_HOST_THREADS", str(max(1, usable_cpus() // world)))

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def allreduce(vals, op):
        if dist is None:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=op)
        return [float(v) for v in t.tolist()]

    from nanopore_amd import job, realign as R
    ctx = R.Context(local_rank)
    job._ctx_pool.setdefault(local_rank, []).insert(0, ctx)  # the job's pipeline uses this context and one more on the same GPU
    env = dict(args=args, ctx=ctx, rank=rank, world=world, dist=dist, coll_dev=coll_dev, sync=sync, allreduce=allreduce, gpu=local_rank)
    if args.ab:
        out = ab_run(env)
    elif args.workload == "c3":
        out = c3_job(env, args.reads or 50000, args.steps, args.warmup, from_files=not args.resident_arrays)
    elif args.workload == "em":
        out = em_step(env)
    else:
        out = resident(env)
        if args.workload == "northstar" and not args.no_also:
            # collective: the strong-scaling job of configs[2]/[3], files -> file, at this N
            # (NPR_BENCH_ALSO_READS: a smaller set for the shared-GPU test hook, where two ranks' scratch must fit one device)
            entry = c3_job(env, int(os.environ.get("NPR_BENCH_ALSO_READS", 50000)), ALSO_STEPS, 1, from_files=True)
            if rank == 0:
                out.setdefault("also", []).append(entry)
    if rank == 0:
        print(json.dumps(out))
    job.close_contexts()
    if dist is not None:
        dist.destroy_process_group()


def ab_run(env):
    """--ab NAME=a,b: the two values step by step in one process, so that what is compared shares the box, its clocks and its neighbours.  One GPU."""
    args, ctx, sync, gpu, coll_dev = (env[k] for k in ("args", "ctx", "sync", "gpu", "coll_dev"))
    from nanopore_amd import _lib, job
    name, _, vals = args.ab.partition("=")
    va, vb = (int(v) for v in vals.split(","))
    pairs = max(1, args.ab_pairs)
    times = {va: [], vb: []}
    dp = {va: [], vb: []}
    if name == "job_overlap":
        n_reads = args.reads or 50000
        h, w, W, tmp, sam, fa = c3_inputs(n_reads, 0, None, True)
        ctxs = job.contexts(gpu, job.WORKERS)
        for c in ctxs:
            c.release_scratch()
            c.set_hmm(h)
        params = make_params(W)
        k = [0]

        def one(v):
            k[0] += 1
            os.environ["NPR_JOB_OVERLAP"] = str(v)
            t0 = time.perf_counter()
            r = job.realign_sam_file(sam, os.path.join(tmp, "ab_%d.sam" % k[0]), fa, params=params, gpu=gpu, set_models=False, coll_device=coll_dev)
            times[v].append((time.perf_counter() - t0) * 1e3)
            dp[v].append(r["timings"]["kernel_ms"])
        one(va), one(vb)
        for v in (va, vb):
            times[v].clear(), dp[v].clear()
        for i in range(pairs):
            first, second = (va, vb) if i % 2 == 0 else (vb, va)  # (the order alternates too: whatever a step leaves behind for the next one is shared out)
            one(first), one(second)
        os.environ.pop("NPR_JOB_OVERLAP", None)
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
        what = "files -> file job of %d reads, NPR_JOB_OVERLAP" % n_reads
    else:
        opt = _lib.OPTIONS[name]
        wl = args.workload if args.workload in ("northstar", "c2", "anchor", "rescore") else "anchor"
        n_reads = args.reads or {"northstar": 24576, "c2": 1000, "anchor": 8192, "rescore": 8192}[wl]
        h, w, W, label = build_workload(wl, n_reads, 0)
        ctx.set_hmm(h)
        batches = {}
        for v in (va, vb):
            ctx.set_option(opt, v)
            batches[v] = ctx.stage_csr(make_params(W), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], guide_start=w.get("guide_start"))
            batches[v].run(), batches[v].finish()

        def one(v):
            t0 = time.perf_counter()
            dp[v].append(batches[v].run())
            batches[v].finish()
            times[v].append((time.perf_counter() - t0) * 1e3)
        sync()
        for i in range(pairs):
            first, second = (va, vb) if i % 2 == 0 else (vb, va)
            one(first), one(second)
        for v in (va, vb):
            batches[v].close()
        what = "%s, context option %s" % (label, name)
    d = np.asarray(times[vb]) - np.asarray(times[va])
    dd = np.asarray(dp[vb]) - np.asarray(dp[va])
    return {"ab": what, "values": [va, vb], "pairs": pairs,
            "step": {str(v): spread(times[v]) for v in (va, vb)}, "dp_launches": {str(v): spread(dp[v]) for v in (va, vb)},
            "paired_difference_ms": {"of": "%d minus %d, step by step" % (vb, va), "mean": float(d.mean()), "std": float(d.std(ddof=1)) if d.size > 1 else 0.0,
                                     "min": float(d.min()), "max": float(d.max()),
                                     "verdict": ("%d is faster" % (vb if d.mean() < 0 else va)) if abs(d.mean()) > 2.0 * d.std(ddof=1) / np.sqrt(d.size) and d.size > 1 else "no difference beyond the spread"},
            "paired_difference_dp_ms": {"mean": float(dd.mean()), "std": float(dd.std(ddof=1)) if dd.size > 1 else 0.0}}


def resident(env):
    args, ctx, rank, world, dist, coll_dev, sync, allreduce = (env[k] for k in ("args", "ctx", "rank", "world", "dist", "coll_dev", "sync", "allreduce"))
    n_reads = args.reads or {"northstar": 24576, "c2": 1000, "anchor": 8192, "rescore": 8192}[args.workload]
    h, w, W, label = build_workload(args.workload, n_reads, rank)
    ctx.set_hmm(h)
    r = timed_resident(ctx, h, w, W, args.steps, args.warmup, sync)
    batch, st, res = r["batch"], r["stats"], r["res"]
    cells = st["cells"]
    elapsed = r["elapsed"]
    if dist is not None:
        elapsed = allreduce([elapsed], dist.ReduceOp.MAX)[0]
        total_cells, total_reads = (int(v) for v in allreduce([cells, n_reads], dist.ReduceOp.SUM))
    else:
        total_cells, total_reads = cells, n_reads

    # the one gather the path has (summary to rank 0), outside the timed steps
    gather_ms = None
    if dist is not None:
        from nanopore_amd import dist as npd
        import torch
        off, ops = batch.ops()
        payload = npd.pack_results(np.arange(n_reads) + rank * n_reads, res["status"], res["score"], off, ops)
        torch.cuda.synchronize()
        tg = time.perf_counter()
        got = npd.gather_to_root(payload, device=coll_dev)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
        if rank == 0:
            status = npd.index_packed_in_input_order(got, n_reads * world)[0]
            assert (status == 0).all()
    if rank != 0:
        batch.close()
        return None
    kms = r["kernel_ms"]
    pairs = int(res["n_pairs"].sum())
    out = {
        "metric": "DP cells/sec (banded pair-HMM realign: forward + backward + posterior per cell)",
        "value": total_cells * args.steps / elapsed,
        "unit": "cells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": label, "reads_per_gpu": n_reads, "band": W, "cells_per_gpu": int(cells),
                   "tasks": int(st["n_tasks"]), "resident_workgroups": int(st["slots"]),
                   "kernel_variant": int(st["kernel_variant"]), "parallelism": "reads sharded x%d" % world},
        "reads_per_s": total_reads * args.steps / elapsed,
        "roofline": roofline_block(cells, pairs, kms, r["class_cells"], gpu_clock_hz()),
        "ok_reads": int((res["status"] == 0).sum()),
        "step": ("npr_batch_run (DP sweep) + npr_batch_finish (the guide's M columns looked up where the pairs lie, summed in fixed point on the device: "
                 "eight bytes per read come back)" if W < 0 else
                 "npr_batch_run (DP sweep) + npr_batch_finish (MEA chain + cigar on the device, ops to the host)"),
        "dp_sweep_only": {"value": cells / (kms * 1e-3), "unit": "cells/s", "ms": kms},
        "finish_ms": r["finish_ms"],
        "gather_ms": gather_ms,
    }
    if world == 1:
        out["from_cold"] = from_cold(ctx, w, r["params"], cells)
        out["from_cold"]["first_create_s"] = r["first_create_s"]
    if world == 1 and not args.no_cpu_baseline:
        cb, ro = cpu_baseline(h, w, W, res["cells"])
        out["cpu_baseline"] = cb
        # the sample doubles as an end-of-run parity probe: GPU cigars vs the fp64 oracle's
        off, ops = batch.ops()
        k = len(ro["ops"])
        same = sum(1 for i in range(k) if np.array_equal(ops[off[i]:off[i + 1]], ro["ops"][i]))
        out["cigar_identical_to_builds_own_fp64_oracle"] = "%d/%d (PARITY UNPINNED: the oracle is this build's restatement, " \
                                                           "the reference binary is absent)" % (same, k)
        if W < 0:  # rescore mode: the cigars are the guides'; what is computed is the score
            out["max_score_difference_to_builds_own_fp64_oracle"] = float(np.max(np.abs(res["score"][:k] - np.asarray(ro["score"])[:k]))) if k else 0.0
    batch.close()
    if world == 1 and args.workload == "northstar" and not args.no_also:
        # the band the drop-in path actually runs (realignSamFile: the reference's own call parameters), driver-visible
        h2, w2, W2, label2 = build_workload("anchor", 8192, 0)
        r2 = timed_resident(ctx, h2, w2, W2, ALSO_STEPS, 1, sync)
        c2, k2 = r2["stats"]["cells"], r2["kernel_ms"]
        out["also"] = [{"workload": label2, "reads": 8192, "value": c2 * ALSO_STEPS / r2["elapsed"], "unit": "cells/s", "steps": ALSO_STEPS,
                        "reads_per_s": 8192 * ALSO_STEPS / r2["elapsed"], "ms_per_step": r2["elapsed"] / ALSO_STEPS * 1e3,
                        "step_spread": spread(r2["step_ms"]), "dp_launch_spread": spread(r2["kernel_ms_all"]),
                        "dp_sweep_only": {"value": c2 / (k2 * 1e-3), "unit": "cells/s", "ms": k2},
                        "roofline": roofline_block(c2, int(r2["res"]["n_pairs"].sum()), k2, r2["class_cells"], gpu_clock_hz()),
                        "ok_reads": int((r2["res"]["status"] == 0).sum())}]
        r2["batch"].close()
        # ... and the posterior consumers' call-site mode (alignmentUncertainty.py:41), finished on the device since round 5
        h3, w3, W3, label3 = build_workload("rescore", 8192, 0)
        r3 = timed_resident(ctx, h3, w3, W3, ALSO_STEPS, 1, sync)
        c3_, k3 = r3["stats"]["cells"], r3["kernel_ms"]
        out["also"].append({"workload": label3, "reads": 8192, "value": c3_ * ALSO_STEPS / r3["elapsed"], "unit": "cells/s", "steps": ALSO_STEPS,
                            "reads_per_s": 8192 * ALSO_STEPS / r3["elapsed"], "ms_per_step": r3["elapsed"] / ALSO_STEPS * 1e3,
                            "step_spread": spread(r3["step_ms"]), "dp_launch_spread": spread(r3["kernel_ms_all"]),
                            "dp_sweep_only": {"value": c3_ / (k3 * 1e-3), "unit": "cells/s", "ms": k3}, "finish_ms": r3["finish_ms"],
                            "roofline": roofline_block(c3_, int(r3["res"]["n_pairs"].sum()), k3, r3["class_cells"], gpu_clock_hz()),
                            "ok_reads": int((r3["res"]["status"] == 0).sum())})
        r3["batch"].close()
        # ... and the trainer's pass (utils.py:509-528: 3 x 100 of these per trained model): the Baum-Welch E-step on the trainer's own band
        em = em_step(env, steps=ALSO_STEPS, warmup=1, cpu=False, reads=6144)
        out["also"].append({"workload": "E-step: " + em["config"]["workload"], "reads": em["config"]["reads_per_gpu"], "value": em["value"], "unit": em["unit"],
                            "steps": em["steps"], "ms_per_step": em["ms_per_step"], "step_spread": em["step_spread"], "roofline": em["roofline"],
                            "class_cells": em["config"]["class_cells"]})
    return out


def c3_inputs(n_reads, rank, dist, from_files):
    """The one read set of configs[2], made by rank 0.  from_files: written as SAM (one local record per read, POS = where
    its window starts on the contig) + FASTA (the 4.6 Mb contig) for every rank to map; else handed to the other ranks as
    arrays through an .npz."""
    from nanopore_amd import synth
    h = load_model()
    tmp = os.path.join(tempfile.gettempdir(), "npr_bench_c3_%d_%d" % (n_reads, os.getuid()))
    sam, fa, npz = (os.path.join(tmp, k) for k in ("reads.sam", "contig.fa", "arrays.npz"))
    keys = ("ref", "ref_off", "ref_index", "read", "read_off", "guide_ops", "guide_off", "guide_start", "interval_len")
    w = None
    if rank == 0:
        os.makedirs(tmp, exist_ok=True)
        w, W = synth.config_c3_shared(h.transitions, h.emissions, n_reads=n_reads)
        if from_files:
            synth.write_workload_files(w, sam, fa, ref_names=["ecoli_like_contig"])
        elif dist is not None:
            np.savez(npz, **{k: w[k] for k in keys})
    if dist is not None:
        dist.barrier()
        if rank != 0 and not from_files:
            z = np.load(npz)
            w = {k: z[k] for k in keys}
        dist.barrier()
    return h, w, 200, tmp, sam, fa


def c3_job(env, n_reads, steps, warmup, from_files):
    import shutil
    from nanopore_amd import job
    ctx, rank, world, dist, coll_dev, sync, allreduce, gpu = (env[k] for k in ("ctx", "rank", "world", "dist", "coll_dev", "sync", "allreduce", "gpu"))
    h, w, W, tmp, sam, fa = c3_inputs(n_reads, rank, dist, from_files)
    ctxs = job.contexts(gpu, job.WORKERS)
    for c in ctxs:
        c.release_scratch()  # as a fresh process would find the device: the resident workloads before this one leave a scratch sized for
        c.set_hmm(h)         # THEIR batches and gigabytes of cached buffers behind (three chunks in flight then do not fit and are halved)
    params = make_params(W)
    step_no = [0]
    out_sam = None

    def one():
        # a fresh output path per step, as a job has: truncating the previous step's 0.6 GB of page cache is not part of it
        nonlocal out_sam
        step_no[0] += 1
        if from_files:
            out_sam = os.path.join(tmp, "realigned_%d.sam" % step_no[0])  # one file, written by all ranks
            return job.realign_sam_file(sam, out_sam, fa, params=params, gpu=gpu, set_models=False, coll_device=coll_dev)
        out_dir = os.path.join(tmp, "out_%d" % step_no[0])
        out_sam = os.path.join(out_dir, "realigned.sam")
        return job.run_job(ctx, params, w, out_dir=out_dir, device=coll_dev)

    last = None
    for _ in range(warmup):
        last = one()
    sync()
    t0 = time.perf_counter()
    tms, step_ms = [], []
    for _ in range(steps):
        ts = time.perf_counter()
        last = one()
        step_ms.append((time.perf_counter() - ts) * 1e3)  # (this rank's; the line's ms_per_step is the maximum over the ranks of the whole region)
        tms.append(last["timings"])
    sync()
    elapsed = time.perf_counter() - t0
    cells_rank = tms[-1]["cells"]
    solo, solo_ms = None, None
    if dist is not None:
        elapsed = allreduce([elapsed], dist.ReduceOp.MAX)[0]
        total_cells = int(allreduce([cells_rank], dist.ReduceOp.SUM)[0])
        if from_files:
            # the SAME job on rank 0 alone (the other ranks wait at the barrier), with the host threads a one-rank launch has:
            # the N = 1 point of the strong-scaling series measured inside the N-rank run, on the same box, so that
            # `speedup_vs_n1` does not depend on a second invocation
            if rank == 0:
                threads_env = os.environ.get("NPR_HOST_THREADS")
                os.environ["NPR_HOST_THREADS"] = str(usable_cpus())

                def alone():
                    step_no[0] += 1
                    return job.realign_sam_file(sam, os.path.join(tmp, "realigned_solo_%d.sam" % step_no[0]), fa, params=params, gpu=gpu,
                                                set_models=False, group=job.SOLO)
                alone()
                solo_ms = []
                for _ in range(steps):
                    ts = time.perf_counter()
                    alone()
                    solo_ms.append((time.perf_counter() - ts) * 1e3)
                solo = float(np.median(solo_ms)) * 1e-3  # (the median: one slow step of seven must not move the series' reference point)
                if threads_env is None:
                    del os.environ["NPR_HOST_THREADS"]
                else:
                    os.environ["NPR_HOST_THREADS"] = threads_env
            sync()
    else:
        total_cells = cells_rank
    if rank != 0:
        return None
    mean = {k: float(np.mean([t[k] for t in tms])) for k in tms[0] if k not in ("cells", "trace")}
    if tms[-1].get("trace"):  # NPR_JOB_TRACE=1: the last step's phases, milliseconds from the step's first event
        base = min(ev[1] for ev in tms[-1]["trace"])
        for ev in sorted(tms[-1]["trace"], key=lambda e: e[1]):
            sys.stderr.write("[job trace] %-7s %8.1f .. %8.1f ms\n" % (ev[0], (ev[1] - base) * 1e3, (ev[2] - base) * 1e3))
    sam_bytes = os.path.getsize(out_sam)
    res = last["results"]
    ok = int((res["status"] == 0).sum())
    in_bytes = (os.path.getsize(sam) + os.path.getsize(fa)) if from_files else 0
    shutil.rmtree(tmp, ignore_errors=True)
    kms = mean["kernel_ms"]
    wall = elapsed / steps
    return {
        "metric": "DP cells/sec (banded pair-HMM realign: forward + backward + posterior per cell), whole job %s" % ("files -> file" if from_files else "from host arrays"),
        "value": total_cells * steps / elapsed,
        "unit": "cells/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": wall * 1e3,
        "step_spread_rank0": spread(step_ms),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[2]/[3]: one set of %d synthetic ~8kb reads on one shared 4.6 Mb contig "
                               "(local records, POS = window start), band 200, blasr_hmm_0, sharded over the ranks into contiguous "
                               "ranges balanced by record length (dist.shard_ranges)" % n_reads,
                   "reads": n_reads, "band": W, "cells": total_cells, "parallelism": "one read set sharded x%d, %d batches in flight per rank" % (world, job.WORKERS),
                   "input_bytes": in_bytes, "output_bytes": sam_bytes},
        "reads_per_s": n_reads * steps / elapsed,
        "ok_reads": ok,
        "step": ("per rank: map SAM + FASTA, parse its shard (native), then chunks of ~6 k reads (5e7 bases) as a pipeline on its GPU (stage = plan + pack + "
                 "H2D + device planner | DP | finish = MEA chain + cigar | fetch + splice of the records' bytes: one thread per phase), blocks written at "
                 "the rank's offset of the one output; per-read results gathered to rank 0 (RCCL) inside the timed region") if from_files else
                "per rank: the same pipeline over arrays resident in host memory, records formatted natively",
        "rank0_phase_seconds": mean,
        "dp_share_of_wall": (kms * 1e-3) / wall,
        "speedup_vs_n1": None if solo is None else solo / wall,
        "n1_ms_per_step": None if solo is None else solo * 1e3,
        "n1_step_spread": None if solo is None else spread(solo_ms),
        "n1_reads_per_s": None if solo is None else n_reads / solo,
        "speedup_note": ("the same files -> file job run by rank 0 ALONE inside this launch (the other ranks wait), all host cores "
                         "to it: ms_per_step(N = 1) / ms_per_step(N = %d); the >= 6x target at N = 8 is read off this field" % world)
                        if solo is not None else "N = 1: this line is the reference point of the series",
        "dp_sweep_only_rank0": {"value": cells_rank / (kms * 1e-3), "unit": "cells/s", "ms": kms},
        "note": "phase seconds are sums over the step's chunks (the phases of different chunks overlap; a DP pass or MEA stage waiting for the "
                "device's shared scratch counts in its phase); dp_share_of_wall = HIP-event time of the DP launches / wall time of a step",
    }


def em_step(env, steps=None, warmup=None, cpu=True, reads=None):
    """Baum-Welch E-step over one resident batch in the trainer's band: a step = npr_batch_expectations (forward with all five
    states kept, backward with the expected transition / emission counts accumulated, one reduction per model slot)."""
    args, ctx, rank, world, dist, sync, allreduce = (env[k] for k in ("args", "ctx", "rank", "world", "dist", "sync", "allreduce"))
    from nanopore_amd import realign as R, synth
    h = load_model()
    steps, warmup = steps or args.steps, args.warmup if warmup is None else warmup
    n_reads = reads or args.reads or 6144
    w = synth.make_workload(1006 + 7919 * rank, n_reads, 4000, h.transitions, h.emissions, flank=0)
    ctx.set_hmm(h)
    P = R.make_params(band_mode=R.BAND_ANCHOR, split_threshold=300, mode=R.MODE_EXPECTATIONS)  # options.optionsToRealign, utils.py:511
    b = ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
    st = b.stats()
    _, class_cells = b.class_stats()
    for _ in range(warmup):
        b.expectations()
    sync()
    t0 = time.perf_counter()
    kms, step_ms = [], []
    for _ in range(steps):
        t1 = time.perf_counter()
        T, E, ll, ms = b.expectations()
        kms.append(ms)
        step_ms.append((time.perf_counter() - t1) * 1e3)
    sync()
    elapsed = time.perf_counter() - t0
    cells = st["cells"]
    if dist is not None:
        elapsed = allreduce([elapsed], dist.ReduceOp.MAX)[0]
        total_cells = int(allreduce([cells], dist.ReduceOp.SUM)[0])
    else:
        total_cells = cells
    if rank != 0:
        b.close()
        return None
    kms = float(np.mean(kms))
    dom = int(np.argmax(class_cells))
    em_bytes = EM_CS_BYTES_PER_CELL if dom == 18 else EM_BYTES_PER_CELL
    achieved = em_bytes * cells / (kms * 1e-3) / 1e9
    tab = kernel_table().get(EM_CLASS_KERNEL.get(dom, ""), {})
    out = {
        "metric": "DP cells/sec (Baum-Welch E-step: forward + backward + expected counts per cell)",
        "value": total_cells * steps / elapsed, "unit": "cells/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "step_spread": spread(step_ms), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "synthetic ~4kb reads in the trainer's band (anchors +- 10, trim 14, splitMatrixBiggerThanThis 300: "
                               "nanopore/analyses/utils.py:511), blasr_hmm_0", "reads_per_gpu": n_reads, "cells_per_gpu": int(cells),
                   "tasks": int(st["n_tasks"]), "class_cells": {EM_CLASS_KERNEL.get(c, CLASS_KERNEL.get(c, str(c))): int(v) for c, v in enumerate(class_cells) if v}},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": (tab["hbm_bytes_per_cell"] * cells / 1e9) if tab.get("hbm_bytes_per_cell") else None,
                     "kernel": EM_CLASS_KERNEL.get(dom, "k_dp_generic<EM>"), "kernel_ms": kms,
                     "kernel_share_of_cells": float(class_cells[dom]) / max(float(np.sum(class_cells)), 1.0),
                     "algorithmic_bytes": ("%g B/cell: the five forward states of every cell (match value 4 B, the other four as 24-bit floats 12 B; exponents per "
                                           "lane and block) stored by the forward sweep and reloaded by the backward sweep" % em_bytes) if dom == 18 else
                                          ("%g B/cell: the five forward states of every cell (match mantissa + exponent 8 B, the other four 16 B) "
                                           "stored by the forward sweep and reloaded by the backward sweep" % em_bytes),
                     "note": "kernel_ms = HIP-event time of ALL E-step launches of the batch (one per kernel class, concurrent)"},
        "loglik": float(ll[0]),
    }
    if world == 1 and cpu and not args.no_cpu_baseline:
        out["cpu_baseline"] = em_cpu_baseline(h, w, P)
    b.close()
    return out


def em_cpu_baseline(h, w, P, budget_s=12.0):
    """orc_expectations_f64 (the fp64 E-step of the oracle: kind "port") over the first reads of the batch, one process
    per core (the oracle's E-step entry point is per segment; there is no OpenMP batch form)."""
    import multiprocessing as mp
    cores = usable_cpus()
    n = len(w["read_off"]) - 1
    k = min(n, 4 * cores)
    global _EM_JOB
    _EM_JOB = (h.transitions, h.emissions, w)  # inherited by the forked workers: no pickling of the batch per task
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_em_cpu_one, range(cores), chunksize=1)  # (warm: library load, first touch)
        t0 = time.time()
        parts = pool.map(_em_cpu_one, range(k), chunksize=1)
        dt = time.time() - t0
    cells = int(sum(parts))
    return {"value": cells / dt, "unit": "cells/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": "first %d reads of the same batch (%d cells): plan (anchors, split 300) + orc_expectations_f64 per segment, fp64 log space, "
                      "%d worker processes, %.1f s" % (k, cells, cores, dt)}


_EM_JOB = None


def _em_cpu_one(i):
    from oracle import oracle as orc
    from nanopore_amd.realign import encode
    T, E, w = _EM_JOB
    oh = orc.make_hmm(T, E)
    P = orc.make_params(band_mode=orc.BAND_ANCHOR, split_threshold=300)
    X = encode(bytes(w["ref"][w["ref_off"][i]:w["ref_off"][i + 1]]))
    Y = encode(bytes(w["read"][w["read_off"][i]:w["read_off"][i + 1]]))
    g = w["guide_ops"][w["guide_off"][i]:w["guide_off"][i + 1]]
    cells = 0
    for seg in orc.plan(len(X), len(Y), g, P):
        r = orc.expectations(oh, X[seg["xs"]:seg["xe"]], Y[seg["ys"]:seg["ye"]], seg["lo"], seg["n"], seg["ragged_start"], seg["ragged_end"])
        assert r["rc"] == 0
        cells += seg["cells"]
    return cells


if __name__ == "__main__":
    main()
