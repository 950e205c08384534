"""tools/finish_breakdown.py [reads] -- where npr_batch_finish goes on the headline batch: the library's own stage times (NPR_TIMING=1 on stderr) over three
passes; run under tools/kstats.sh for the kernels' times.  Bring-up tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["NPR_TIMING"] = "1"
import time
import bench
from nanopore_amd import realign as R
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
h, w, W, label = bench.build_workload("northstar", n, 0)
ctx = R.Context(0); ctx.set_hmm(h)
b = ctx.stage_csr(bench.make_params(W), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], guide_start=w.get("guide_start"))
for i in range(3):
    ms = b.run()
    t0 = time.perf_counter()
    b.finish()
    sys.stderr.write("==== pass %d: dp %.1f ms, finish %.1f ms\n" % (i, ms, (time.perf_counter() - t0) * 1e3))
b.close()
