"""Bring-up check of k_dp_rs (row-scaled arithmetic): GPU results against the oracle's row-scaled mirror on random fixed-band
and anchor-band cases, bit for bit, with the mismatch counts printed instead of asserted.  Not part of the product."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from helpers import MODEL_DIR, oracle_hmm, orc, random_pair  # noqa: E402
from nanopore_amd import realign as R  # noqa: E402
from nanopore_amd.hmm import Hmm  # noqa: E402

ctx = R.Context(0)
ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + "/blasr_hmm_0.txt"))
h = oracle_hmm()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
bad_total = 0
for kw, n, lmin, lmax, indel, mi in ((dict(band_mode=1, fixed_width=40), 24, 5, 300, 0.12, 4),
                                     (dict(band_mode=1, fixed_width=100), 12, 200, 1500, 0.2, 30),
                                     (dict(band_mode=1, fixed_width=200), 12, 300, 2500, 0.2, 40),
                                     (dict(band_mode=1, fixed_width=400), 6, 400, 1500, 0.2, 40),
                                     (dict(band_mode=0, diagonal_expansion=10, constraint_trim=2, split_threshold=12), 16, 50, 500, 0.2, 40),
                                     (dict(band_mode=0, diagonal_expansion=60, constraint_trim=3, split_threshold=3000), 8, 300, 3000, 0.15, 20),
                                     # the reference's own band: unanchored rectangles up to 3000 cells across -> the stripe kernel
                                     (dict(band_mode=0, diagonal_expansion=10, constraint_trim=14, split_threshold=3000), 6, 1500, 6000, 0.2, 40),
                                     (dict(band_mode=0, diagonal_expansion=10, constraint_trim=14, split_threshold=700, max_pairs_per_base=40), 4, 1500, 4000, 0.25, 60),
                                     (dict(band_mode=1, fixed_width=700), 4, 400, 1500, 0.2, 40)):
    cases = [random_pair(rng, int(rng.integers(lmin, lmax + 1)), indel=indel, max_indel=mi) for _ in range(n)]
    refs = [bytes(b"ACGT"[c] for c in X) for X, _, _ in cases]
    reads = [bytes(b"ACGT"[c] for c in Y) for _, Y, _ in cases]
    out = ctx.realign(R.make_params(**kw), refs, reads, [g for _, _, g in cases], want_pairs=True)
    okw = dict(kw)
    okw.pop("max_pairs_per_base", None)
    P = orc.make_params(**okw)
    bad = 0
    narith = 0
    for (X, Y, ops), g in zip(cases, out):
        m = orc.realign_read(h, P, X, Y, ops, precision=1, seg_arith=g["seg_arith"])
        narith += sum(g["seg_arith"])
        gp = {(int(a), int(b)): float(c) for a, b, c in zip(g["x"], g["y"], g["p"])}
        mp = {(int(a), int(b)): float(np.float32(c)) for a, b, c in zip(m["px"], m["py"], m["pp"])}
        ok = g["status"] == 0 and gp.keys() == mp.keys() and all(gp[k] == mp[k] for k in gp) and g["ops"] == m["ops"] and \
            abs(g["loglik"] - m["total_ll"]) < 1e-9 * max(1.0, abs(m["total_ll"]))
        if not ok:
            bad += 1
            common = set(gp) & set(mp)
            print("  MISMATCH len", len(X), "status", g["status"], "pairs gpu/mirror/common", len(gp), len(mp), len(common),
                  "max |dp|", max([abs(gp[k] - mp[k]) for k in common] or [0.0]), "ll", g["loglik"], m["total_ll"], "bwd", g["loglik_bwd"],
                  "arith", g["seg_arith"][:6])
    bad_total += bad
    print(kw, ":", n, "reads,", narith, "row-scaled segments,", bad, "mismatching reads", flush=True)
print("TOTAL MISMATCHES", bad_total)
