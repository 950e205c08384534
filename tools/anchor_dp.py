"""tools/anchor_dp.py tag... -- DP launch time of library variants (NPR_LIB) on the reference's own band (8192 reads, anchors +- 10, trim 14, split 3000:
k_dp_tile) and on the E-step of the trainer's band.  Bring-up tool."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
from nanopore_amd import realign as R, synth
from nanopore_amd.hmm import Hmm
h = Hmm.loadHmm(os.path.join(%(root)r, "nanopore_amd", "mappers", "blasr_hmm_0.txt"))
ctx = R.Context(0); ctx.set_hmm(h)
out = {}
w = synth.make_workload(1004, 8192, 8000, h.transitions, h.emissions)
b = ctx.stage_csr(R.make_params(band_mode=R.BAND_ANCHOR, max_pairs_per_base=24), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
out["anchor_ms"] = round(min(b.run() for _ in range(3)), 1); b.finish(); out["score"] = float(b.results()["score"].mean()); b.close()
w = synth.make_workload(1006, 2048, 4000, h.transitions, h.emissions, flank=0)
b = ctx.stage_csr(R.make_params(band_mode=R.BAND_ANCHOR, split_threshold=300, mode=R.MODE_EXPECTATIONS), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
ms = [b.expectations()[3] for _ in range(3)]
out["em_ms"] = round(min(ms), 1); b.close()
print(json.dumps(out))
'''
for tag in sys.argv[1:]:
    lib = os.path.join(ROOT, "nanopore_amd", "libnprealign.so" if tag == "default" else "libnprealign_%s.so" % tag)
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=dict(os.environ, NPR_LIB=lib), capture_output=True, text=True)
    print(tag, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-500:], flush=True)
