import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import numpy as np
from nanopore_amd import realign as R, synth
from nanopore_amd.hmm import Hmm
h = Hmm.loadHmm(os.path.join(ROOT, 'nanopore_amd', 'mappers', 'blasr_hmm_0.txt'))
n, L, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
w = synth.make_workload(7, n, L, h.transitions, h.emissions, flank=0)
ctx = R.Context(0); ctx.set_hmm(h)
P = R.make_params(band_mode=1, fixed_width=W) if W > 0 else R.make_params(band_mode=0, max_pairs_per_base=24)
for rep in range(4):
    if rep == 3: os.environ['NPR_TIMING'] = '1'
    t0 = time.perf_counter()
    b = ctx.stage_csr(P, w['ref'], w['ref_off'], w['read'], w['read_off'], w['guide_ops'], w['guide_off'])
    t1 = time.perf_counter()
    print('create %.1f ms' % ((t1 - t0) * 1e3), flush=True)
    b.close()
