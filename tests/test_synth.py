import numpy as np

from helpers import load_model_arrays
from nanopore_amd import synth


def _spans(g):
    return int(g[g[:, 0] != 1, 1].sum()), int(g[g[:, 0] != 2, 1].sum())


def test_workload_is_seeded_and_guides_are_global():
    T, E, _ = load_model_arrays()
    w, W = synth.config_c2(T, E, n_reads=60)
    w2, _ = synth.config_c2(T, E, n_reads=60)
    assert W == 100 and (w["read"] == w2["read"]).all() and (w["guide_ops"] == w2["guide_ops"]).all()
    n = len(w["ref_off"]) - 1
    moved = 0
    for i in range(n):
        g = w["guide_ops"][w["guide_off"][i]:w["guide_off"][i + 1]]
        assert _spans(g) == (w["ref_off"][i + 1] - w["ref_off"][i], w["read_off"][i + 1] - w["read_off"][i])
        assert (g[:, 1] > 0).all() and (g[1:, 0] != g[:-1, 0]).all()
        t = w["true_runs"][w["true_off"][i]:w["true_off"][i + 1]]
        moved += int(len(g) != len(t) or (g != t).any())
    assert moved > n // 2                       # the guide is a degraded version of the truth
    rl = w["read_off"][1:] - w["read_off"][:-1]
    assert 800 < rl.mean() < 1250


def test_north_star_shape_has_50kb_slices():
    T, E, _ = load_model_arrays()
    w, W = synth.config_north_star(T, E, n_reads=8, windowed=False)
    assert W == 200 and ((w["ref_off"][1:] - w["ref_off"][:-1]) == 50000).all()
    g = w["guide_ops"][w["guide_off"][0]:w["guide_off"][1]]
    assert _spans(g)[0] == 50000 and w["guide_start"] is None
    # default: the guide keeps its own span and carries the coordinates of its window inside the slice
    v, _ = synth.config_north_star(T, E, n_reads=8)
    assert ((v["ref_off"][1:] - v["ref_off"][:-1]) == 50000).all() and np.array_equal(v["ref"], w["ref"])
    for i in range(8):
        g = v["guide_ops"][v["guide_off"][i]:v["guide_off"][i + 1]]
        sx, sy = _spans(g)
        assert sx == v["interval_len"][i] and sy == v["read_off"][i + 1] - v["read_off"][i]
        assert v["guide_start"][i, 0] == v["lead"][i] and v["guide_start"][i, 1] == 0
        assert v["guide_start"][i, 0] + sx <= 50000
