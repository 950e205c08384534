"""tools/variant_bench.py tag [tag ...] -- DP launch time of library variants (make -C nanopore_amd/csrc variant TAG=.. EXTRA=..)
on the north-star batch and on the reference's own band, each in its own process (NPR_LIB picks the library), plus how many
cigars of a small sample equal the fp64 oracle's (a variant that changes the arithmetic -- the renormalisation rule -- is not
bit-identical to the default mirror; this only says that it is still the same alignment).  Bring-up tool."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
from nanopore_amd import realign as R, synth
from nanopore_amd.hmm import Hmm
h = Hmm.loadHmm(os.path.join(%(root)r, "nanopore_amd", "mappers", "blasr_hmm_0.txt"))
ctx = R.Context(0); ctx.set_hmm(h)
out = {}
w, W = synth.config_north_star(h.transitions, h.emissions, n_reads=12288, seed=1003)
b = ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], guide_start=w.get("guide_start"))
ms = [b.run() for _ in range(4)]
b.finish()
res = b.results(); off, ops = b.ops()
out["northstar_ms"] = min(ms[1:]); out["northstar_cells_per_s"] = b.stats()["cells"] / min(ms[1:]) * 1e3; out["ok"] = int((res["status"] == 0).sum())
from helpers import orc
oh = orc.make_hmm(h.transitions, h.emissions)
P = orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=W)
k = 48
lead, ilen = w["guide_start"][:k, 0], w["interval_len"][:k]
X = R.encode(b"".join(bytes(w["ref"][w["ref_off"][i] + lead[i]:w["ref_off"][i] + lead[i] + ilen[i]]) for i in range(k)))
x_off = np.concatenate([[0], np.cumsum(ilen)]).astype(np.int64)
Y = R.encode(bytes(w["read"][:w["read_off"][k]]))
for prec, key in ((0, "same_as_fp64"), (1, "same_as_default_mirror")):
    r = orc.realign_batch(oh, P, X, x_off, Y, w["read_off"][:k + 1], w["guide_ops"][:w["guide_off"][k]], w["guide_off"][:k + 1], precision=prec, threads=16, native=True)
    out[key] = "%%d/%%d" %% (sum(int(np.array_equal(ops[off[i]:off[i + 1]], r["ops"][i])) for i in range(k)), k)
    if prec == 0:
        out["max_score_diff_vs_fp64"] = float(np.abs(res["score"][:k] - r["score"]).max())
b.close()
w2 = synth.make_workload(1004, 8192, 8000, h.transitions, h.emissions)
b = ctx.stage_csr(R.make_params(band_mode=R.BAND_ANCHOR, max_pairs_per_base=24), w2["ref"], w2["ref_off"], w2["read"], w2["read_off"], w2["guide_ops"], w2["guide_off"])
ms = [b.run() for _ in range(3)]
out["anchor_ms"] = min(ms[1:]); out["anchor_cells_per_s"] = b.stats()["cells"] / min(ms[1:]) * 1e3
b.close()
print(json.dumps(out))
'''

for tag in sys.argv[1:]:
    lib = os.path.join(ROOT, "nanopore_amd", "libnprealign.so" if tag == "default" else "libnprealign_%s.so" % tag)
    env = dict(os.environ, NPR_LIB=lib)
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True)
    line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-400:]
    print(tag, line, flush=True)
