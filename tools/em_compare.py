"""The E-step of one batch on the three kernel families (stripes / workgroup frames / generic), counts compared.  Bring-up tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from nanopore_amd import _lib, realign as R, synth
from nanopore_amd.hmm import Hmm
h = Hmm.loadHmm(os.path.join(ROOT, 'nanopore_amd', 'mappers', 'blasr_hmm_0.txt'))
n, L, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
w = synth.make_workload(11, n, L, h.transitions, h.emissions, flank=0, length_sigma=0.4, len_min=300, len_max=4 * L)
ctx = R.Context(0); ctx.set_hmm(h)
P = (R.make_params(band_mode=1, fixed_width=W, mode=R.MODE_EXPECTATIONS) if W > 0 else
     R.make_params(band_mode=0, split_threshold=300, mode=R.MODE_EXPECTATIONS))
got = {}
for name, env in (('stripes', {}), ('stripes_percell', {'em_tile': 1}), ('frames', {'no_tile': 1}), ('generic', {'em_generic': 1})):  # context options (include/nprealign.h)
    for k, v in env.items(): ctx.set_option(_lib.OPTIONS[k], v)
    b = ctx.stage_csr(P, w['ref'], w['ref_off'], w['read'], w['read_off'], w['guide_ops'], w['guide_off'])
    tasks, _ = b.class_stats()
    T, E, ll, ms = b.expectations()
    T2, E2, ll2, ms2 = b.expectations()
    b.close()
    for k in env: ctx.set_option(_lib.OPTIONS[k], 0)
    got[name] = (T[0], E[0], ll[0])
    print(name, 'classes', np.nonzero(tasks)[0].tolist(), 'kernel %.1f ms' % ms2, 'll %.6f' % ll[0], 'repeat dT %.2e' % (np.abs(T2[0] - T[0]).max() / T[0].sum()), flush=True)
s = got['frames'][0].sum()
for name in ('stripes', 'stripes_percell'):
    print(name, 'vs frames: dT %.2e dE %.2e dll %.3e' % (np.abs(got[name][0] - got['frames'][0]).max() / s, np.abs(got[name][1] - got['frames'][1]).max() / s, got[name][2] - got['frames'][2]))
print('T stripes', got['stripes'][0].round(1).tolist())
print('T percell', got['stripes_percell'][0].round(1).tolist())
s = got['generic'][0].sum()
for name in ('stripes', 'stripes_percell', 'frames'):
    print(name, 'vs generic: dT %.2e dE %.2e dll %.3e' % (np.abs(got[name][0] - got['generic'][0]).max() / s, np.abs(got[name][1] - got['generic'][1]).max() / s, got[name][2] - got['generic'][2]))
