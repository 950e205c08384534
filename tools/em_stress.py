"""Randomised stress of the stripe E-step (k_dp_tile_cs<.., EM>) against the fp64 oracle and against k_em_tile: random pairs, band widths
and models (flat gap emissions, gap emissions by base, short-gap switches, scaled transition rows).  python tools/em_stress.py [seconds] [seed].
Bring-up tool; the committed tests run fixed cases of this."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from nanopore_amd import _lib, realign as R
from nanopore_amd.hmm import Hmm
from helpers import MODEL_DIR, orc, random_pair
import test_gpu_em as TE

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = R.Context(0)
t0 = time.time(); n = 0; worst = [0.0, 0.0, 0.0]
while time.time() - t0 < budget:
    hm = Hmm.loadHmm(os.path.join(MODEL_DIR, "blasr_hmm_0.txt"))
    T = np.array(hm.transitions).reshape(5, 5).copy(); E = np.array(hm.emissions).reshape(5, 4, 4).copy()
    kind = int(rng.integers(0, 4))
    if kind == 1:
        g = rng.dirichlet(np.ones(4) * 4) / 4
        E[1], E[3] = g[:, None] * np.ones((4, 4)), g[::-1][:, None] * np.ones((4, 4))
        E[2], E[4] = g[None, :] * np.ones((4, 4)), g[::-1][None, :] * np.ones((4, 4))
    if kind == 2:
        a, b = rng.random(2) * 0.05
        T[1, 2], T[2, 1] = a, b; T[1, 1] -= a; T[2, 2] -= b
    if kind == 3:   # other gap-open / extend probabilities
        o = rng.random() * 0.1 + 0.02
        T[0] = [1 - 2 * o - 0.02, o, o, 0.01, 0.01]
        e = rng.random() * 0.5 + 0.2
        T[1] = [1 - e, e, 0, 0, 0]; T[2] = [1 - e, 0, e, 0, 0]
    hm.transitions, hm.emissions = [float(v) for v in T.reshape(-1)], [float(v) for v in E.reshape(-1)]
    ctx.set_hmm(hm)
    W = int(rng.integers(270, 1600))
    cases = [random_pair(rng, int(rng.integers(300, 2200)), indel=float(rng.random() * 0.3), max_indel=int(rng.integers(1, 150))) for _ in range(int(rng.integers(2, 8)))]
    if rng.random() < 0.5:
        cases[0][0][5:9] = 4
    kw = dict(band_mode=1, fixed_width=W) if rng.random() < 0.6 else dict(band_mode=0, diagonal_expansion=int(rng.integers(140, 400)), constraint_trim=int(rng.integers(0, 20)), split_threshold=int(rng.integers(200, 3000)))
    got = {}
    for mode in (0, 1):
        ctx.set_option(_lib.OPTIONS["em_tile"], mode)
        b = ctx.stage(R.make_params(mode=R.MODE_EXPECTATIONS, **kw), [TE._ascii(X) for X, _, _ in cases], [TE._ascii(Y) for _, Y, _ in cases], [g for _, _, g in cases])
        tasks, _ = b.class_stats()
        got[mode] = (b.expectations(), tasks.copy())
        b.close()
    ctx.set_option(_lib.OPTIONS["em_tile"], 0)
    if got[0][1][18] == 0:
        continue
    wantT, wantE, wantll = TE._oracle_expectations(hm.transitions, hm.emissions, cases, orc.make_params(**kw))
    scale = wantT.sum()
    (Tg, Eg, ll, _), _ = got[0]
    (Tp, Ep, llp, _), _ = got[1]
    d = [np.abs(Tg[0] - wantT).max() / scale, np.abs(Eg[0] - wantE).max() / scale, max(np.abs(Tg[0] - Tp[0]).max(), np.abs(Eg[0] - Ep[0]).max()) / scale]
    worst = [max(a, c) for a, c in zip(worst, d)]
    assert d[0] < 2e-5 and d[1] < 2e-5 and d[2] < 1e-5 and abs(ll[0] - wantll) < 2e-6 * abs(wantll), (kind, kw, d, ll[0], wantll)
    n += 1
print("em_stress: %d batches in %.0f s; worst |T - oracle| %.2e, |E - oracle| %.2e, |stripes - k_em_tile| %.2e (of the total count)" % (n, time.time() - t0, *worst))
