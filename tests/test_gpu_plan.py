"""npr_batch_create expands band rows, frame schedules (control words), stripe tables and generic row offsets ON THE
DEVICE from each segment's plan points (nanopore_amd/csrc/npr_plan.hip); the host planner -- the one tests/test_host_logic.py
pins against the oracle -- recomputes every task and npr_batch_plan_check compares entry by entry."""
import numpy as np
import pytest

from helpers import MODEL_DIR, load_model_arrays, random_pair

pytestmark = pytest.mark.gpu


def _ascii(codes):
    return bytes(b"ACGT"[c] for c in codes)


def test_device_plans_equal_host_plans_on_random_guides(gpu_ctx, monkeypatch):
    from nanopore_amd import realign as R
    rng = np.random.default_rng(909)
    for it in range(24):
        cases = [random_pair(rng, int(rng.integers(1, 1500)), indel=rng.random() * 0.3, max_indel=int(rng.integers(1, 80)))
                 for _ in range(24)]
        kw = dict(band_mode=it % 2, diagonal_expansion=int(rng.integers(0, 8)) * 2, constraint_trim=int(rng.integers(0, 16)),
                  split_threshold=int(rng.choice([0, 5, 40, 300, 3000])), fixed_width=int(rng.choice([2, 9, 40, 100, 200, 420, 900])))
        for env in ({}, {"no_tile": 1}, {"kernel": 1}):   # (context options: include/nprealign.h NPR_OPT_NO_TILE, NPR_OPT_KERNEL)
            with gpu_ctx.options(**env):
                b = gpu_ctx.stage(R.make_params(**kw), [_ascii(X) for X, _, _ in cases], [_ascii(Y) for _, Y, _ in cases],
                                  [g for _, _, g in cases])
                assert b.plan_check() == 0, (kw, env)
                b.close()


def test_device_plans_of_the_named_workloads(gpu_ctx):
    from nanopore_amd import realign as R, synth
    from nanopore_amd.hmm import Hmm
    T, E, _ = load_model_arrays()
    gpu_ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + "/blasr_hmm_0.txt"))
    w, W = synth.config_north_star(T, E, n_reads=48)
    b = gpu_ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w["ref"], w["ref_off"], w["read"], w["read_off"],
                          w["guide_ops"], w["guide_off"], guide_start=w["guide_start"])
    assert b.plan_check() == 0 and b.stats()["kernel_variant"] == 1
    b.close()
    w = synth.make_workload(1004, 24, 8000, T, E)
    b = gpu_ctx.stage_csr(R.make_params(band_mode=R.BAND_ANCHOR), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
    assert b.plan_check() == 0 and b.stats()["kernel_variant"] == 2
    b.close()
    # invalid reads next to valid ones: no tasks for them, the others unaffected
    refs, reads, guides = [b"ACGTACGTAC", b"ACGT", b"ACGTACGT"], [b"ACGTACGTAC", b"ACGT", b"ACGTACGT"], [[(0, 10)], [(0, 3)], [(0, 8)]]
    b = gpu_ctx.stage(R.make_params(band_mode=R.BAND_FIXED, fixed_width=10), refs, reads, guides)
    assert b.plan_check() == 0 and b.stats()["n_tasks"] == 2
    b.run(), b.finish()
    assert list(b.results()["status"]) == [0, -1, 0]
    b.close()
