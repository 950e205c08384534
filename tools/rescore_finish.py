"""tools/rescore_finish.py -- npr_batch_finish of the posterior consumers' modes next to the realign mode's on the same batch (bench.py's
rescore workload: 8192 x ~8 kb reads, anchors +- 10, trim 14, split 100): milliseconds per call, device stage and host stage
(NPR_OPT_HOST_MEA = 1).  Bring-up tool."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from nanopore_amd import realign as R, synth, _lib
from nanopore_amd.hmm import Hmm
h = Hmm.loadHmm(os.path.join(ROOT, "nanopore_amd", "mappers", "blasr_hmm_0.txt"))
ctx = R.Context(0); ctx.set_hmm(h)
w = synth.make_workload(1006, int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 8000, h.transitions, h.emissions, jitter=0)
out = {}
for name, mode in (("realign", R.MODE_REALIGN), ("rescore", R.MODE_RESCORE_ORIGINAL), ("all_posteriors", R.MODE_ALL_POSTERIORS)):
    for where in ("device", "host"):
        ctx.set_option(_lib.OPTIONS["host_mea"], int(where == "host"))
        print(name, where, file=sys.stderr, flush=True)
        P = R.make_params(band_mode=R.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=100, mode=mode, max_pairs_per_base=24)
        b = ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
        ms = []
        for _ in range(4):
            k = b.run(); print(' run', k, file=sys.stderr, flush=True); t0 = time.perf_counter(); b.finish(); ms.append((time.perf_counter() - t0) * 1e3); print(' finish', ms[-1], file=sys.stderr, flush=True)
        res = b.results()
        out["%s_%s" % (name, where)] = dict(finish_ms=round(min(ms), 2), dp_ms=round(k, 2), ok=int((res["status"] == 0).sum()), mean_score=float(res["score"].mean()))
        b.close()
print(json.dumps(out, indent=1))
