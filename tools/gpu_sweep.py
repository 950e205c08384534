"""Bring-up sweep (not the contract bench): kernel ms for the north-star-like batch under env knobs."""
import os, sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from nanopore_amd import _lib, realign as R, synth
from nanopore_amd.hmm import Hmm
h = Hmm.loadHmm(os.path.join(ROOT, 'nanopore_amd', 'mappers', 'blasr_hmm_0.txt'))
n = int(sys.argv[1]); L = int(sys.argv[2]); W = int(sys.argv[3])  # W = 0: the reference's anchor band (expansion 10, trim 14, split 3000)
w = synth.make_workload(7, n, L, h.transitions, h.emissions, flank=0)
ctx = R.Context(0); ctx.set_hmm(h)
for wpc in sys.argv[4:]:
    ctx.set_option(_lib.OPTIONS['waves_per_cu'], 0 if wpc == 'auto' else int(wpc))  # NPR_OPT_WAVES_PER_CU
    t0 = time.time()
    P = R.make_params(band_mode=1, fixed_width=W) if W > 0 else R.make_params(band_mode=0)
    b = ctx.stage_csr(P, w['ref'], w['ref_off'], w['read'], w['read_off'], w['guide_ops'], w['guide_off'])
    st = b.stats(); ts = time.time() - t0
    ms = min(b.run() for _ in range(3))
    t0 = time.time(); b.finish(); tf = time.time() - t0
    print('waves/CU %3s slots %5d tasks %d maxw %d  %.2f ms  %.3e cells/s  %.0f reads/s (stage %.2fs finish %.2fs)' % (wpc, st['slots'], st['n_tasks'], st['max_width'], ms, st['cells'] / ms * 1e3, n / ms * 1e3, ts, tf), flush=True)
    b.close()
