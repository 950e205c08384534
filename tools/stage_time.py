"""tools/stage_time.py -- staging (npr_batch_create) of a 1/8 shard of configs[3] and of the headline batch with NPR_TIMING=1: the planner's
laps on stderr.  Bring-up tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["NPR_TIMING"] = "1"
from nanopore_amd import realign as R, synth
from nanopore_amd.hmm import Hmm
h = Hmm.loadHmm(os.path.join(ROOT, "nanopore_amd", "mappers", "blasr_hmm_0.txt"))
ctx = R.Context(0); ctx.set_hmm(h)
for name, (w, W) in (("shard 6250", synth.config_c3_shared(h.transitions, h.emissions, n_reads=6250)), ("headline 24576", synth.config_north_star(h.transitions, h.emissions, n_reads=24576, seed=1003))):
    for rep in range(3):
        sys.stderr.write("---- %s rep %d\n" % (name, rep))
        t0 = time.perf_counter()
        b = ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], ref_index=w.get("ref_index"), guide_start=w.get("guide_start"))
        sys.stderr.write("total %.1f ms\n" % ((time.perf_counter() - t0) * 1e3))
        b.close()
