import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from nanopore_amd import realign as R
from nanopore_amd.hmm import Hmm
from helpers import random_pair
h = Hmm.loadHmm('/root/repo/nanopore_amd/mappers/blasr_hmm_0.txt')
def run(L, W, n, seed, indel=0.1, max_indel=5):
    rng = np.random.default_rng(seed)
    cases = [random_pair(rng, L, indel=indel, max_indel=max_indel) for _ in range(n)]
    refs = [bytes(b"ACGT"[c] for c in X) for X, Y, o in cases]; reads = [bytes(b"ACGT"[c] for c in Y) for X, Y, o in cases]
    out = {}
    for mode in ("stair", "generic"):
        if mode == "generic": os.environ["NPR_EM_GENERIC"] = "1"
        else: os.environ.pop("NPR_EM_GENERIC", None)
        ctx = R.Context(0); ctx.set_hmm(h)
        b = ctx.stage(R.make_params(band_mode=1, fixed_width=W, mode=R.MODE_EXPECTATIONS), refs, reads, [o for X, Y, o in cases])
        T, E, ll, ms = b.expectations()
        out[mode] = (T[0].copy(), E[0].copy(), ll[0])
        b.close(); ctx.close()
    a, g = out["stair"], out["generic"]
    dT = np.abs(a[0] - g[0]); dE = np.abs(a[1] - g[1])
    print("L %d W %d n %d: ll %.6f %.6f  max dT %.4g (of %.4g) at %d  max dE %.4g" % (L, W, n, a[2], g[2], dT.max(), g[0].sum(), dT.argmax(), dE.max()))
    if dT.max() > 1e-3 * g[0].sum(): print(np.round(a[0] - g[0], 4).reshape(5, 5))
for L, W, n in ((30, 20, 1), (200, 40, 1), (200, 40, 8), (600, 100, 4), (600, 200, 4), (300, 20, 4)):
    run(L, W, n, 5)
run(400, 60, 4, 6, indel=0.3)
for sd in range(3):
    run(300, 120, 1, 10 + sd, indel=0.2, max_indel=30)
    run(300, 40, 1, 20 + sd, indel=0.2, max_indel=30)
