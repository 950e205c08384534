"""tools/tile_rs_time.py [reads] -- the reference's own band (anchors +- 10, trim 14, split 3000) on k_dp_tile and on k_dp_tile_rs (NPR_OPT_TILE_RS):
DP launch times, how many tasks ran again without their range certificate (NPR_TIMING=1 on stderr), whether the results agree.  Bring-up tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["NPR_TIMING"] = "1"
import numpy as np
from nanopore_amd import _lib, realign as R, synth
from nanopore_amd.hmm import Hmm
h = Hmm.loadHmm(os.path.join(ROOT, "nanopore_amd", "mappers", "blasr_hmm_0.txt"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
w = synth.make_workload(1004, n, 8000, h.transitions, h.emissions)
ctx = R.Context(0); ctx.set_hmm(h)
P = R.make_params(band_mode=R.BAND_ANCHOR, max_pairs_per_base=24)
out = {}
for name, opt in (("tile", 0), ("tile_rs", 1), ("tile", 0), ("tile_rs", 1)):
    ctx.set_option(_lib.OPTIONS["tile_rs"], opt)
    b = ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
    sys.stderr.write("==== %s\n" % name)
    ms = [b.run() for _ in range(2)]
    b.finish()
    res = b.results()
    out[name] = (res["loglik"].copy(), res["score"].copy())
    sys.stderr.write("==== %s dp ms %s ok %d\n" % (name, [round(m, 1) for m in ms], int((res["status"] == 0).sum())))
    b.close()
print("loglik equal", np.array_equal(out["tile"][0], out["tile_rs"][0]), "score equal", np.array_equal(out["tile"][1], out["tile_rs"][1]))
