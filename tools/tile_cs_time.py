"""tools/tile_cs_time.py [reads] -- the reference's own band (anchors +- 10, trim 14, split 3000) on k_dp_tile and on k_dp_tile_cs (NPR_OPT_TILE_RS):
DP launch times, how many tasks ran again without their range certificate (NPR_TIMING=1 on stderr), whether the results agree: log-likelihoods,
scores, cigars and posterior pairs, bit for bit.  Bring-up tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["NPR_TIMING"] = "1"
import numpy as np
from nanopore_amd import _lib, realign as R, synth
from nanopore_amd.hmm import Hmm
h = Hmm.loadHmm(os.path.join(ROOT, "nanopore_amd", "mappers", "blasr_hmm_0.txt"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
length = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
w = synth.make_workload(1004, n, length, h.transitions, h.emissions)
ctx = R.Context(0); ctx.set_hmm(h)
P = R.make_params(band_mode=R.BAND_ANCHOR, max_pairs_per_base=24)
out = {}
for name, opt in (("tile", 2), ("tile_cs", 0)) * reps:
    ctx.set_option(_lib.OPTIONS["tile_rs"], opt)
    b = ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
    sys.stderr.write("==== %s\n" % name)
    ms = [b.run() for _ in range(2)]
    b.finish()
    res = b.results()
    off, ops = b.ops()
    out[name] = (res["loglik"].copy(), res["loglik_bwd"].copy(), res["score"].copy(), res["status"].copy(), off.copy(), ops.copy()) + tuple(np.array(x).copy() for x in (b.pairs() if n <= 2048 else ()))
    sys.stderr.write("==== %s dp ms %s ok %d\n" % (name, [round(m, 1) for m in ms], int((res["status"] == 0).sum())))
    print(name, "dp ms", [round(m, 1) for m in ms], "ok", int((res["status"] == 0).sum()), "of", n, flush=True)
    b.close()
names = ("loglik", "loglik_bwd", "score", "status", "ops_off", "ops", "pair_off", "px", "py", "pp")
for i, (x, y) in enumerate(zip(out["tile"], out["tile_cs"])):
    same = x.shape == y.shape and np.array_equal(x, y)
    print(names[i], "equal" if same else "DIFFER", "" if same or x.shape != y.shape else "at %d of %d" % (int((x != y).sum()), x.size))
