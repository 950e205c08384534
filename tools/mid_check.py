"""tools/mid_check.py -- bring-up: k_dp_mid_rs (NPR_OPT_PAIR = 3) against k_dp_rs (NPR_OPT_PAIR = 1) on the same batches: posterior pairs, totals
and cigars bit for bit (the lists sorted: the two kernels fill them in different orders); then the DP launch of both on a 1/8 shard of
configs[3], on configs[1] and on the headline batch."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from helpers import random_pair, MODEL_DIR
from nanopore_amd import realign as R, synth, _lib
from nanopore_amd.hmm import Hmm

h = Hmm.loadHmm(MODEL_DIR + '/blasr_hmm_0.txt')
ctx = R.Context(0); ctx.set_hmm(h)

def run(opt, params, refs, reads, guides):
    ctx.set_option(_lib.OPTIONS["pair"], opt)
    return ctx.realign(params, refs, reads, guides, want_pairs=True)

def compare(tag, n, lmin, lmax, kw, indel=0.12, max_indel=4, seed=5):
    rng = np.random.default_rng(seed)
    refs, reads, guides = [], [], []
    for _ in range(n):
        X, Y, ops = random_pair(rng, int(rng.integers(lmin, lmax + 1)), indel=indel, max_indel=max_indel)
        refs.append(bytes(b"ACGT"[c] for c in X)); reads.append(bytes(b"ACGT"[c] for c in Y)); guides.append(ops)
    P = R.make_params(**kw)
    a = run(1, P, refs, reads, guides)
    b = run(3, P, refs, reads, guides)
    bad = 0
    for i, (u, v) in enumerate(zip(a, b)):
        ku = np.lexsort((u["y"], u["x"])); kv = np.lexsort((v["y"], v["x"]))
        same = (len(ku) == len(kv) and np.array_equal(u["x"][ku], v["x"][kv]) and np.array_equal(u["y"][ku], v["y"][kv])
                and np.array_equal(u["p"][ku].view(np.uint32), v["p"][kv].view(np.uint32)) and u["ops"] == v["ops"] and u["loglik"] == v["loglik"]
                and u["loglik_bwd"] == v["loglik_bwd"] and u["status"] == v["status"] == 0 and u["seg_arith"] == v["seg_arith"])
        if not same:
            bad += 1
            if bad <= 3:
                print("  DIFF read", i, "len", len(reads[i]), "pairs", len(ku), len(kv), "status", u["status"], v["status"], "ll", u["loglik"], v["loglik"], u["loglik_bwd"], v["loglik_bwd"], "arith", u["seg_arith"], v["seg_arith"])
                if len(ku) == len(kv):
                    dx = np.flatnonzero((u["x"][ku] != v["x"][kv]) | (u["y"][ku] != v["y"][kv]) | (u["p"][ku].view(np.uint32) != v["p"][kv].view(np.uint32)))
                    print("   first differing", dx[:5], [(u["x"][ku][j], u["y"][ku][j], u["p"][ku][j], v["x"][kv][j], v["y"][kv][j], v["p"][kv][j]) for j in dx[:3]])
    print(tag, "reads", n, "differing", bad, flush=True)
    return bad

bad = 0
bad += compare("R1 small", 48, 30, 400, dict(band_mode=1, fixed_width=40))
bad += compare("R1 tiny/odd", 64, 31, 70, dict(band_mode=1, fixed_width=40), seed=6)
bad += compare("R2 drift", 16, 300, 1500, dict(band_mode=1, fixed_width=200), indel=0.2, max_indel=40, seed=7)
bad += compare("R4 drift", 8, 400, 900, dict(band_mode=1, fixed_width=400), indel=0.2, max_indel=30, seed=8)
bad += compare("R2 long", 6, 4000, 9000, dict(band_mode=1, fixed_width=200), seed=9)
bad += compare("R1 anchors", 16, 200, 800, dict(band_mode=0), seed=10)
print("TOTAL differing", bad, flush=True)
if "--time" in sys.argv:
    out = {}
    for name, opt in (("rs", 1), ("mid", 3), ("default", 0)):
        ctx.set_option(_lib.OPTIONS["pair"], opt)
        w, W = synth.config_c3_shared(h.transitions, h.emissions, n_reads=6250)
        b = ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], ref_index=w["ref_index"], guide_start=w["guide_start"])
        out[name + "_shard_ms"] = round(min(b.run() for _ in range(4)), 2); b.close()
        w, W = synth.config_c2(h.transitions, h.emissions)
        b = ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], guide_start=w.get("guide_start"))
        out[name + "_c2_ms"] = round(min(b.run() for _ in range(6)), 3); b.close()
        w, W = synth.config_north_star(h.transitions, h.emissions, n_reads=24576, seed=1003)
        b = ctx.stage_csr(R.make_params(band_mode=R.BAND_FIXED, fixed_width=W), w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"], guide_start=w.get("guide_start"))
        out[name + "_ns_ms"] = round(min(b.run() for _ in range(3)), 2); b.close()
        print(json.dumps(out), flush=True)
