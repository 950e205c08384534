import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from nanopore_amd import realign as R, synth
from nanopore_amd.hmm import Hmm
ROOT='/root/repo'
h = Hmm.loadHmm(os.path.join(ROOT, 'nanopore_amd', 'mappers', 'blasr_hmm_0.txt'))
n=int(sys.argv[1]); L=int(sys.argv[2]); W=int(sys.argv[3])
w = synth.make_workload(7, n, L, h.transitions, h.emissions, flank=0)
ctx = R.Context(0); ctx.set_hmm(h)
from nanopore_amd import _lib
for kv in filter(None, os.environ.get('NPR_OPTS', '').split(',')):  # e.g. NPR_OPTS=class_min=3,em_tile=1 (context options, include/nprealign.h)
    k, v = kv.split('='); ctx.set_option(_lib.OPTIONS[k], int(v))
P = R.make_params(band_mode=1, fixed_width=W, mode=R.MODE_EXPECTATIONS) if W > 0 else R.make_params(band_mode=0, split_threshold=300, mode=R.MODE_EXPECTATIONS)  # the trainer's own options (utils.py:511)
b = ctx.stage_csr(P, w['ref'], w['ref_off'], w['read'], w['read_off'], w['guide_ops'], w['guide_off'])
st = b.stats()
print('class cells', [int(v) for v in b.class_stats()[1]], flush=True)
ms = min(b.run() for _ in range(2))
t0=time.time(); T,E,ll,kms = b.expectations(); t1=time.time()
reps = sorted(b.expectations()[3] for _ in range(int(os.environ.get('EM_REPS', '7'))))
kms2 = reps[len(reps) // 2]
print('E-step kernel ms: median %.2f min %.2f max %.2f of %d' % (kms2, reps[0], reps[-1], len(reps)))
print('n %d L %d W %d cells %.3e: realign kernel %.1f ms (%.2e cells/s); E-step kernel %.1f ms (%.2e cells/s), wall %.2f s' % (n,L,W,st['cells'],ms,st['cells']/ms*1e3,kms2,st['cells']/kms2*1e3,t1-t0))
