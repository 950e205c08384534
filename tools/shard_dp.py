"""tools/shard_dp.py tag[:option=value,...]... -- DP launch time of library variants (NPR_LIB) and context options (nanopore_amd/_lib.py
OPTIONS, e.g. default:pair=2) on a 1/8 shard of configs[3] (6250 reads: the launch lasts as long as its longest read), on config 2
and on the headline batch, each in its own process.  Bring-up tool."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
from nanopore_amd import realign as R, synth
from nanopore_amd.hmm import Hmm
h = Hmm.loadHmm(os.path.join(%(root)r, "nanopore_amd", "mappers", "blasr_hmm_0.txt"))
ctx = R.Context(0); ctx.set_hmm(h)
from nanopore_amd import _lib
for kv in filter(None, %(opts)r.split(",")):
    k, v = kv.split("=")
    if k.startswith("NPR_"): os.environ[k] = v
    else: ctx.set_option(_lib.OPTIONS[k], int(v))
out = {}
w, W = synth.config_c3_shared(h.transitions, h.emissions, n_reads=6250)
b = ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], ref_index=w["ref_index"], guide_start=w["guide_start"])
out["shard_ms"] = round(min(b.run() for _ in range(4)), 2); b.close()
w, W = synth.config_c2(h.transitions, h.emissions)
b = ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], guide_start=w.get("guide_start"))
out["c2_ms"] = round(min(b.run() for _ in range(6)), 3); b.close()
w, W = synth.config_north_star(h.transitions, h.emissions, n_reads=24576, seed=1003)
b = ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], guide_start=w.get("guide_start"))
out["ns_ms"] = round(min(b.run() for _ in range(3)), 2); b.close()
print(json.dumps(out))
'''
for arg in sys.argv[1:]:
    tag, _, opts = arg.partition(":")
    lib = os.path.join(ROOT, "nanopore_amd", "libnprealign.so" if tag == "default" else "libnprealign_%s.so" % tag)
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "opts": opts}], env=dict(os.environ, NPR_LIB=lib), capture_output=True, text=True)
    print(arg, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-500:], flush=True)
