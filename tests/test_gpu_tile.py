"""k_dp_tile (the wide-band register kernel on column stripes, nanopore_amd/csrc/npr_kernel_tile.hip) on the GPU, through
the C ABI: bit-exact against the oracle's fp32 mirror and within 1e-4 of the fp64 log-space oracle (test_gpu_parity's two
bars), bit-identical to the kernels it replaces (k_dp_wide / k_dp_generic under NPR_OPT_NO_TILE), and independent of how many
wavefronts share a task."""
import numpy as np
import pytest

from nanopore_amd import _lib

from helpers import MODEL_DIR, load_model_arrays, orc, seg_arith_of
from test_gpu_parity import _run_case

pytestmark = pytest.mark.gpu


def _codes(buf):
    from nanopore_amd.realign import encode
    return encode(bytes(buf))


def test_wide_fixed_bands_match_the_oracle(gpu_ctx):
    """Constant-width bands from just over one wavefront's frame (256 slots) to several stripes: every read against the
    fp32 mirror (bit-exact) and the fp64 oracle (1e-4)."""
    rng = np.random.default_rng(31)
    _run_case(gpu_ctx, rng, 4, 300, 600, dict(band_mode=1, fixed_width=300), indel=0.2, max_indel=20)
    _run_case(gpu_ctx, rng, 3, 500, 900, dict(band_mode=1, fixed_width=700), indel=0.2, max_indel=30)
    _run_case(gpu_ctx, rng, 2, 1300, 1600, dict(band_mode=1, fixed_width=2600, max_pairs_per_base=12), indel=0.2, max_indel=30)


def test_anchor_bands_match_the_oracle(gpu_ctx):
    """The reference's band shape: unanchored rectangles joined by narrow stripes, with and without matrix splits
    (ragged ends start / end in the long-gap states)."""
    rng = np.random.default_rng(32)
    _run_case(gpu_ctx, rng, 6, 400, 1500, dict(band_mode=0, diagonal_expansion=10, constraint_trim=14, split_threshold=3000,
                                              max_pairs_per_base=12), indel=0.3, max_indel=8)
    _run_case(gpu_ctx, rng, 6, 400, 1500, dict(band_mode=0, diagonal_expansion=10, constraint_trim=14, split_threshold=300,
                                              max_pairs_per_base=40), indel=0.3, max_indel=8)


def test_tile_kernel_takes_the_reference_band_and_agrees_with_the_other_kernels(gpu_ctx, monkeypatch):
    """Reads from the shipped nanopore model under the reference's call parameters (nanopore/analyses/utils.py:587): the
    stripe kernel takes every band wider than one wavefront's frame; results are those of k_dp_wide / k_dp_generic bit for
    bit, whatever the number of wavefronts per task."""
    from nanopore_amd import realign as R, synth
    from nanopore_amd.hmm import Hmm
    T, E, _ = load_model_arrays()
    w = synth.make_workload(1007, 48, 3000, T, E, flank=0, length_sigma=0.5, len_min=300, len_max=9000)
    gpu_ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + "/blasr_hmm_0.txt"))

    def run(P):
        b = gpu_ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
        tasks, cells = b.class_stats()
        b.run(), b.finish()
        out = (b.results(), b.ops(), b.pairs(), tasks, cells, b.stats(), seg_arith_of(b))
        b.close()
        return out

    def same(a, c):
        assert np.array_equal(a[0]["cells"], c[0]["cells"]) and np.array_equal(a[0]["status"], c[0]["status"])
        assert np.array_equal(a[0]["loglik"], c[0]["loglik"]) and np.array_equal(a[0]["loglik_bwd"], c[0]["loglik_bwd"])
        assert np.array_equal(a[0]["score"], c[0]["score"])
        assert np.array_equal(a[1][0], c[1][0]) and np.array_equal(a[1][1], c[1][1])
        for k in range(4):
            assert np.array_equal(a[2][k], c[2][k])

    for P in (R.make_params(band_mode=R.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=3000),
              R.make_params(band_mode=R.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=700,
                            max_pairs_per_base=40)):
        gpu_ctx.set_option(_lib.OPTIONS["tile_rs"], 2)  # k_dp_tile, one exponent per cell: what the other kernels are compared with
        ref = run(P)
        res, tasks, cells, st = ref[0], ref[3], ref[4], ref[5]
        assert (res["status"] == 0).all() and np.abs(res["loglik"] - res["loglik_bwd"]).max() < 1e-2
        assert tasks[3:11].sum() == 0 and tasks[11] > 0 and cells[11] > 0.9 * cells.sum() and st["kernel_variant"] == 2
        for nw in ("1", "3", "4"):
            gpu_ctx.set_option(_lib.OPTIONS["tile_waves"], int(nw))
            same(ref, run(P))
        gpu_ctx.set_option(_lib.OPTIONS["tile_waves"], 0)
        gpu_ctx.set_option(_lib.OPTIONS["no_tile"], 1)
        old = run(P)
        gpu_ctx.set_option(_lib.OPTIONS["no_tile"], 0)
        assert old[3][11] == 0 and old[3][3:11].sum() == tasks[11]
        same(ref, old)
        # the stripes in column-scaled arithmetic (k_dp_tile_cs, the default since round 6: one exponent per lane of a stripe, neighbour cells
        # handed over with their lane's exponent, tasks without a range certificate run again in k_dp_tile): the same bits, whatever the number
        # of wavefronts per task
        gpu_ctx.set_option(_lib.OPTIONS["tile_rs"], 0)
        for nw in ("0", "1", "3"):
            gpu_ctx.set_option(_lib.OPTIONS["tile_waves"], int(nw))
            cs = run(P)
            assert cs[3][18] > 0 and cs[3][11] == 0 and ref[3][11] == cs[3][18]
            same(ref, cs)
        gpu_ctx.set_option(_lib.OPTIONS["tile_waves"], 0)

    # two reads against the oracle's fp32 mirror
    h = orc.make_hmm(T, E)
    PO = orc.make_params(band_mode=orc.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=3000)
    got = run(R.make_params(band_mode=R.BAND_ANCHOR))
    res, (off, ops), (poff, px, py, pp), arith = got[0], got[1], got[2], got[6]
    order_by_cells = np.argsort(res["cells"])
    for i in (int(order_by_cells[len(order_by_cells) // 2]), int(order_by_cells[5])):
        X = _codes(w["ref"][w["ref_off"][i]:w["ref_off"][i + 1]])
        Y = _codes(w["read"][w["read_off"][i]:w["read_off"][i + 1]])
        g = [tuple(int(v) for v in r) for r in w["guide_ops"][w["guide_off"][i]:w["guide_off"][i + 1]]]
        m = orc.realign_read(h, PO, X, Y, g, precision=1, seg_arith=arith(i))
        assert m["cells"] == res["cells"][i]
        assert [tuple(int(v) for v in r) for r in ops[off[i]:off[i + 1]]] == m["ops"]
        order = np.lexsort((m["py"], m["px"]))
        assert np.array_equal(pp[poff[i]:poff[i + 1]], m["pp"].astype(np.float32)[order])


def test_repeated_launches_on_poisoned_scratch_give_identical_pairs(gpu_ctx, monkeypatch):
    """Every launch finds its scratch and its buffers filled with a poison byte (NPR_POISON) instead of what the previous,
    identical launch left there, several kernel classes run side by side, and every result must equal the first one's: a
    kernel that reads a row before it is written, or from a wrong address, shows up here instead of once in a hundred
    launches on one box in three.  (This caught the row offset of k_dp_stair held in a short-lived SGPR as the buffer
    instruction's scalar offset: tools/stress_tile.py)"""
    from nanopore_amd import realign as R, synth
    from nanopore_amd.hmm import Hmm
    T, E, _ = load_model_arrays()
    gpu_ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + "/blasr_hmm_0.txt"))
    monkeypatch.setenv("NPR_POISON", "0x3f")
    cases = ((synth.make_workload(1007, 48, 3000, T, E, flank=0, length_sigma=0.5, len_min=300, len_max=9000),
              R.make_params(band_mode=R.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=3000), 120),
             (synth.make_workload(1008, 256, 1500, T, E, flank=0, length_sigma=0.4, len_min=200, len_max=5000),
              R.make_params(band_mode=R.BAND_FIXED, fixed_width=100), 60),
             (synth.make_workload(1009, 96, 2000, T, E, flank=0, length_sigma=0.4, len_min=200, len_max=5000),
              R.make_params(band_mode=R.BAND_FIXED, fixed_width=300), 60),
             # band 200: the north-star class (k_dp_mid_rs<2>, packed control words), next to a few narrow and wide stragglers
             (synth.make_workload(1010, 192, 2500, T, E, flank=0, length_sigma=0.6, len_min=100, len_max=8000),
              R.make_params(band_mode=R.BAND_FIXED, fixed_width=200), 80),
             # ... and the same class launched side by side with others: anchors +- 60 with 3 trimmed columns give bands of
             # 64-126 cells for most reads of this workload (class 1), wider ones where the indels cluster (class 2), one narrow
             (synth.make_workload(1011, 128, 3000, T, E, flank=0, length_sigma=0.5, len_min=300, len_max=9000),
              R.make_params(band_mode=R.BAND_ANCHOR, diagonal_expansion=60, constraint_trim=3, split_threshold=3000), 120))
    seen_classes = set()
    for w, P, reps in cases:
        first = None
        for rep in range(reps):
            b = gpu_ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
            if rep == 0:
                tasks, _ = b.class_stats()
                seen_classes.add(tuple(int(c) for c in np.nonzero(tasks)[0]))
            b.run(), b.finish()
            out = (b.results(), b.pairs(), b.ops())
            b.close()
            if first is None:
                first = out
                assert (out[0]["status"] == 0).all()
                continue
            for key in ("status", "loglik", "loglik_bwd", "score"):
                assert np.array_equal(out[0][key], first[0][key]), (rep, key)
            assert all(np.array_equal(a, c) for a, c in zip(out[1], first[1])), rep
            assert all(np.array_equal(a, c) for a, c in zip(out[2], first[2])), rep
    # the north-star class (13: class 1's frame in row-scaled arithmetic on two wavefronts, k_dp_mid_rs<2>) ran alone and next to other classes
    assert (13,) in seen_classes and any(13 in c and len(c) > 1 for c in seen_classes), seen_classes
