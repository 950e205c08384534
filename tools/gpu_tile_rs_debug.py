"""Bring-up: the reference-band workload of tests/test_gpu_tile.py with NPR_TILE_RS=1 (k_dp_tile_rs) and through the default
path (k_dp_tile): class statistics, status histogram, first differences.  Not part of the product."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from helpers import MODEL_DIR, load_model_arrays  # noqa: E402
from nanopore_amd import realign as R, synth  # noqa: E402
from nanopore_amd.hmm import Hmm  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
T, E, _ = load_model_arrays()
w = synth.make_workload(1007, n, L, T, E, flank=0, length_sigma=0.5, len_min=300, len_max=3 * L)
ctx = R.Context(0)
ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + "/blasr_hmm_0.txt"))
P = R.make_params(band_mode=R.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=3000)


def run():
    b = ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
    tasks, cells = b.class_stats()
    ms = b.run()
    b.finish()
    out = (b.results(), b.ops(), b.pairs(), tasks, ms)
    b.close()
    return out


os.environ["NPR_TILE_RS"] = "1"
a = run()
print("tile_rs: classes", {int(c): int(v) for c, v in enumerate(a[3]) if v}, "ms %.2f" % a[4], "status", dict(zip(*np.unique(a[0]["status"], return_counts=True))))
del os.environ["NPR_TILE_RS"]
c = run()
print("cell:    classes", {int(k): int(v) for k, v in enumerate(c[3]) if v}, "ms %.2f" % c[4], "status", dict(zip(*np.unique(c[0]["status"], return_counts=True))))
bad = [i for i in range(n) if a[0]["status"][i] != c[0]["status"][i] or a[0]["loglik"][i] != c[0]["loglik"][i] or a[0]["loglik_bwd"][i] != c[0]["loglik_bwd"][i]]
print("reads with another status / total:", len(bad), bad[:10])
for i in bad[:5]:
    print("  read", i, "len", int(w["read_off"][i + 1] - w["read_off"][i]), "status", a[0]["status"][i], c[0]["status"][i], "ll", a[0]["loglik"][i], c[0]["loglik"][i],
          "bwd", a[0]["loglik_bwd"][i], c[0]["loglik_bwd"][i], "pairs", a[0]["n_pairs"][i], c[0]["n_pairs"][i])
same_pairs = all(np.array_equal(x, y) for x, y in zip(a[2], c[2]))
print("pairs identical:", same_pairs, "ops identical:", np.array_equal(a[1][1], c[1][1]))
