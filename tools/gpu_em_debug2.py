import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from nanopore_amd import realign as R
from nanopore_amd.hmm import Hmm
h = Hmm.loadHmm('/root/repo/nanopore_amd/mappers/blasr_hmm_0.txt')
rng = np.random.default_rng(3)
def run(ops, W, label):
    lX = sum(n for o, n in ops if o in (0, 2)); lY = sum(n for o, n in ops if o in (0, 1))
    X = rng.integers(0, 4, lX); Y = rng.integers(0, 4, lY)
    # make matches agree
    x = y = 0
    for o, n in ops:
        if o == 0:
            Y[y:y + n] = X[x:x + n]; x += n; y += n
        elif o == 2: x += n
        else: y += n
    refs = [bytes(b"ACGT"[c] for c in X)]; reads = [bytes(b"ACGT"[c] for c in Y)]
    out = {}
    for mode in ("stair", "generic"):
        if mode == "generic": os.environ["NPR_EM_GENERIC"] = "1"
        else: os.environ.pop("NPR_EM_GENERIC", None)
        ctx = R.Context(0); ctx.set_hmm(h)
        P = R.make_params(band_mode=1, fixed_width=W)
        b = ctx.stage(P, refs, reads, [ops])
        T, E, ll, ms = b.expectations()
        out[mode] = T[0].copy()
        b.close(); ctx.close()
    sch = R.frame_schedule(R.make_params(band_mode=1, fixed_width=W), lX, lY, ops, 64 if W < 126 else 128, 1 if W < 126 else 2)
    d = np.abs(out["stair"] - out["generic"])
    print("%-28s W %3d: max dT %.4g of %.4g; rebases +%d -%d" % (label, W, d.max(), out["generic"].sum(), (sch["rebase"] > 0).sum(), (sch["rebase"] < 0).sum()))
for W in (50, 124):
    run([(0, 150)], W, "matches only")
    run([(0, 100), (2, 60), (0, 100)], W, "one 60-base deletion")
    run([(0, 100), (1, 60), (0, 100)], W, "one 60-base insertion")
    run([(0, 60), (2, 45), (0, 60), (1, 50), (0, 60)], W, "deletion then insertion")
    run([(0, 60), (1, 45), (0, 60), (2, 50), (0, 60)], W, "insertion then deletion")
    run([(0, 10), (1, 30), (0, 10), (2, 30), (0, 10), (1, 5), (2, 5), (0, 20)], W, "short blocks")
