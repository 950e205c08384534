"""Kernel-time experiment runner (bring-up only): python tools/gpu_exp.py LIB n L W [flankslice]
Loads an alternative build of libnprealign.so (tools/exp/*.so, built with -DNPR_EXP=k) and times npr_batch_run only;
results of such builds are garbage by design and never read."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import nanopore_amd._lib as L_
if sys.argv[1] != 'default':
    L_.LIB_PATH = os.path.join(ROOT, sys.argv[1])
from nanopore_amd import realign as R, synth
from nanopore_amd.hmm import Hmm
h = Hmm.loadHmm(os.path.join(ROOT, 'nanopore_amd', 'mappers', 'blasr_hmm_0.txt'))
n = int(sys.argv[2]); L = int(sys.argv[3]); W = int(sys.argv[4]); sl = int(sys.argv[5]) if len(sys.argv) > 5 else 0
w = synth.make_workload(7, n, L, h.transitions, h.emissions, flank=0, ref_slice_len=(sl or None))
ctx = R.Context(0); ctx.set_hmm(h)
P = R.make_params(band_mode=1, fixed_width=W) if W > 0 else R.make_params(band_mode=0)
b = ctx.stage_csr(P, w['ref'], w['ref_off'], w['read'], w['read_off'], w['guide_ops'], w['guide_off'])
st = b.stats()
ms = min(b.run() for _ in range(3))
print('%-28s n %d L %d W %d slice %d: %.2f ms  %.3e cells/s' % (sys.argv[1], n, L, W, sl, ms, st['cells'] / ms * 1e3), flush=True)
