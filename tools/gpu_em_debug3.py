import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from nanopore_amd import realign as R
from nanopore_amd.hmm import Hmm
from helpers import random_pair
h = Hmm.loadHmm('/root/repo/nanopore_amd/mappers/blasr_hmm_0.txt')
rng = np.random.default_rng(77)
_ = [random_pair(rng, int(rng.integers(60, 900)), indel=0.25, max_indel=60) for _ in range(12)]
W = 124
cases = [random_pair(rng, int(rng.integers(200, 1200)), indel=0.25, max_indel=60) for _ in range(12)]
for i, (X, Y, g) in enumerate(cases):
    out = {}
    for mode in ("stair", "generic"):
        if mode == "generic": os.environ["NPR_EM_GENERIC"] = "1"
        else: os.environ.pop("NPR_EM_GENERIC", None)
        ctx = R.Context(0); ctx.set_hmm(h)
        b = ctx.stage(R.make_params(band_mode=1, fixed_width=W), [bytes(b"ACGT"[c] for c in X)], [bytes(b"ACGT"[c] for c in Y)], [g])
        T, E, ll, ms = b.expectations()
        out[mode] = T[0].copy()
        b.close(); ctx.close()
    d = np.abs(out["stair"] - out["generic"])
    sch = R.frame_schedule(R.make_params(band_mode=1, fixed_width=W), len(X), len(Y), g, 64, 1)
    print("read %2d lX %4d lY %4d ops %3d: max dT %.4g of %.4g  rebases +%d -%d" % (i, len(X), len(Y), len(g), d.max(), out["generic"].sum(),
          (sch["rebase"] > 0).sum() if sch else -1, (sch["rebase"] < 0).sum() if sch else -1))
    if d.max() > 1e-2:
        print("   guide:", [tuple(int(v) for v in o) for o in g])
        print(np.round(out["stair"] - out["generic"], 3).reshape(5, 5))
