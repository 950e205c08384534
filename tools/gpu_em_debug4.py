import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from nanopore_amd import realign as R
from nanopore_amd.hmm import Hmm
from helpers import random_pair
h = Hmm.loadHmm('/root/repo/nanopore_amd/mappers/blasr_hmm_0.txt')
rng = np.random.default_rng(77)
ctx = R.Context(0); ctx.set_hmm(h)
for W, lens in ((50, (60, 900)), (124, (200, 1200))):
    cases = [random_pair(rng, int(rng.integers(*lens)), indel=0.25, max_indel=60) for _ in range(12)]
    refs = [bytes(b"ACGT"[c] for c in X) for X, Y, g in cases]; reads = [bytes(b"ACGT"[c] for c in Y) for X, Y, g in cases]
    for order in (("register", "generic"), ("generic", "register"), ("register", "register")):
        res = []
        for mode in order:
            if mode == "generic": os.environ["NPR_EM_GENERIC"] = "1"
            else: os.environ.pop("NPR_EM_GENERIC", None)
            b = ctx.stage(R.make_params(band_mode=1, fixed_width=W), refs, reads, [g for X, Y, g in cases])
            T = b.expectations()[0][0].copy(); b.close()
            res.append(T[24])
        print(W, order, res)
