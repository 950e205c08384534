"""Committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py).

CPU leg: the oracle still reproduces them (pins the oracle against silent drift).
GPU leg: the HIP path through the C ABI reproduces them -- bit-exact against the fp32-mirror outputs (posterior
bits, cigar), within 1e-4 of the fp64 outputs."""
import glob
import os

import numpy as np
import pytest

from helpers import ROOT, load_model_arrays, orc

CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
ASCII = np.frombuffer(b"ACGTN", dtype=np.uint8)


def _params(z, make):
    bm, de, ct, st, fw, mode = (int(v) for v in z["params"])
    gg, mg = (float(v) for v in z["gammas"])
    return make(band_mode=bm, diagonal_expansion=de, constraint_trim=ct, split_threshold=st, fixed_width=fw,
                gap_gamma=gg, match_gamma=mg, mode=mode)


def test_fixture_set_is_present():
    assert len(CASES) >= 7


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_reproduces_golden(path):
    z = np.load(path)
    T, E, _ = load_model_arrays(str(z["model"]))
    h = orc.make_hmm(T, E)
    guide = [tuple(int(v) for v in r) for r in z["guide"]]
    r64 = orc.realign_read(h, _params(z, orc.make_params), z["X"], z["Y"], guide, precision=0)
    r32 = orc.realign_read(h, _params(z, orc.make_params), z["X"], z["Y"], guide, precision=1)
    assert r64["cells"] == int(z["cells"])
    assert r64["total_ll"] == pytest.approx(float(z["f64_total_ll"]), rel=1e-13)
    assert np.array_equal(np.array(r64["ops"], dtype=np.int32).reshape(-1, 2), z["f64_ops"].reshape(-1, 2))
    assert np.array_equal(r64["px"], z["f64_px"]) and np.array_equal(r64["py"], z["f64_py"])
    assert np.abs(r64["pp"] - z["f64_pp"]).max() < 1e-12
    assert np.array_equal(np.array(r32["ops"], dtype=np.int32).reshape(-1, 2), z["f32_ops"].reshape(-1, 2))
    assert np.array_equal(r32["pp"].astype(np.float32), z["f32_pp"])          # the mirror is bit-reproducible
    assert r32["score"] == float(z["f32_score"])
    # the row-scaled restatement of the same recurrences (realign_oracle_rs.c: the arithmetic of the one-wavefront kernels)
    # lands on the same floats: scaling by powers of two is exact, so the two fp32 arithmetics part only where a cell falls
    # out of fp32's range relative to its row -- nowhere near a posterior that is reported
    rrs = orc.realign_read(h, _params(z, orc.make_params), z["X"], z["Y"], guide, precision=1, seg_arith=[1] * 4096)
    assert np.array_equal(np.array(rrs["ops"], dtype=np.int32).reshape(-1, 2), z["f32_ops"].reshape(-1, 2))
    key = lambda r: np.lexsort((r["py"], r["px"]))  # noqa: E731
    a, b = key(rrs), key(r32)
    assert np.array_equal(rrs["px"][a], r32["px"][b]) and np.array_equal(rrs["py"][a], r32["py"][b])
    assert np.array_equal(rrs["pp"][a].astype(np.float32), r32["pp"][b].astype(np.float32))
    assert rrs["score"] == float(z["f32_score"]) and rrs["total_ll"] == r32["total_ll"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_gpu_reproduces_golden(path, gpu_ctx):
    from nanopore_amd import realign as R
    from nanopore_amd.hmm import Hmm
    z = np.load(path)
    gpu_ctx.set_hmm(Hmm.loadHmm(os.path.join(ROOT, "nanopore_amd", "mappers", str(z["model"]))))
    guide = [tuple(int(v) for v in r) for r in z["guide"]]
    out = gpu_ctx.realign(_params(z, R.make_params), [ASCII[z["X"]].tobytes()], [ASCII[z["Y"]].tobytes()], [guide],
                          want_pairs=True)[0]
    assert out["status"] == 0 and out["cells"] == int(z["cells"])
    order = np.lexsort((z["f32_py"], z["f32_px"]))
    assert np.array_equal(out["x"], z["f32_px"][order]) and np.array_equal(out["y"], z["f32_py"][order])
    assert np.array_equal(out["p"], z["f32_pp"][order]), "posterior bits differ from the golden fp32 vector"
    assert np.array_equal(np.array(out["ops"], dtype=np.int32).reshape(-1, 2), z["f32_ops"].reshape(-1, 2))
    assert out["score"] == pytest.approx(float(z["f32_score"]), abs=1e-15)
    # and the fp64 vectors within the stated tolerance
    assert out["loglik"] == pytest.approx(float(z["f64_total_ll"]), rel=2e-6)
    d64 = {(int(a), int(b)): float(c) for a, b, c in zip(z["f64_px"], z["f64_py"], z["f64_pp"])}
    for a, b, c in zip(out["x"], out["y"], out["p"]):
        assert abs(d64.get((int(a), int(b)), 0.01) - float(c)) < 1e-4
    assert np.array_equal(np.array(out["ops"], dtype=np.int32).reshape(-1, 2), z["f64_ops"].reshape(-1, 2))
