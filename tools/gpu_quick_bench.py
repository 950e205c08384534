"""Bring-up throughput probe (not the contract bench): synthetic fixed-band batch, kernel ms."""
import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from helpers import *
from nanopore_amd import realign as R
from nanopore_amd.hmm import Hmm
nreads, L, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = R.Context(0)
ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + '/blasr_hmm_0.txt'))
rng = np.random.default_rng(1)
refs, reads, guides = [], [], []
base = [random_pair(rng, L) for _ in range(min(nreads, 64))]
for i in range(nreads):
    X, Y, ops = base[i % len(base)]
    refs.append(bytes(b"ACGT"[c] for c in X)); reads.append(bytes(b"ACGT"[c] for c in Y)); guides.append(ops)
t0 = time.time()
b = ctx.stage(R.make_params(band_mode=1, fixed_width=W), refs, reads, guides)
st = b.stats(); print('stage s', time.time() - t0, st, flush=True)
for it in range(3):
    ms = b.run()
    print('run ms %.3f  cells/s %.3e  GB/s@40B %.1f' % (ms, st['cells'] / ms * 1e3, st['cells'] * 40 / ms * 1e-6), flush=True)
t0 = time.time(); b.finish(); print('finish s', time.time() - t0)
r = b.results(); print('status ok', (r['status'] == 0).all(), 'score mean', r['score'].mean(), 'pairs/base', r['n_pairs'].sum() / (nreads * L))
