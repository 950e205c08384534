"""tools/rs_variants.py tag[:waves_per_cu] ... -- DP launch time of library variants of k_dp_rs (make -C nanopore_amd/csrc
variant TAG=.. EXTRA=..; "default" = the built library) on the north-star batch and on config 2, each in its own process.
Bring-up tool."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np
from nanopore_amd import realign as R, synth
from nanopore_amd.hmm import Hmm
h = Hmm.loadHmm(os.path.join(%(root)r, "nanopore_amd", "mappers", "blasr_hmm_0.txt"))
ctx = R.Context(0); ctx.set_hmm(h)
if os.environ.get("RS_WAVES_PER_CU"): ctx.set_option(10, int(os.environ["RS_WAVES_PER_CU"]))  # NPR_OPT_WAVES_PER_CU
out = {}
w, W = synth.config_north_star(h.transitions, h.emissions, n_reads=int(os.environ.get("RS_NS_READS", "12288")), seed=1003)
b = ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], guide_start=w.get("guide_start"))
ms = [b.run() for _ in range(5)]
b.finish(); res = b.results()
out["ns_ms"] = round(min(ms[1:]), 2); out["ns_cells_per_s"] = "%%.3e" %% (b.stats()["cells"] / min(ms[1:]) * 1e3); out["ok"] = int((res["status"] == 0).sum())
out["score"] = float(res["score"].mean())
b.close()
if os.environ.get("RS_ONLY") == "ns":
    print(json.dumps(out)); sys.exit(0)
w = synth.make_workload(77, 49152, 2500, h.transitions, h.emissions, flank=0)
b = ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=200), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
ms = [b.run() for _ in range(4)]
out["many_ms"] = round(min(ms[1:]), 2); out["many_cells_per_s"] = "%%.3e" %% (b.stats()["cells"] / min(ms[1:]) * 1e3)
b.close()
w, W = synth.config_c2(h.transitions, h.emissions)
b = ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], guide_start=w.get("guide_start"))
ms = [b.run() for _ in range(6)]
out["c2_ms"] = round(min(ms[1:]), 3)
b.close()
print(json.dumps(out))
'''
for arg in sys.argv[1:]:
    tag, _, wpc = arg.partition(":")
    lib = os.path.join(ROOT, "nanopore_amd", "libnprealign.so" if tag == "default" else "libnprealign_%s.so" % tag)
    env = dict(os.environ); env.setdefault("NPR_LIB", lib)
    if wpc:
        env["RS_WAVES_PER_CU"] = wpc
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True)
    line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-600:]
    print(arg, line, flush=True)
