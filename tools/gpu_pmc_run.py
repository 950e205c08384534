"""One staged batch, a few DP launches: the target of rocprofv3 --pmc passes."""
import os, sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from nanopore_amd import realign as R, synth
from nanopore_amd.hmm import Hmm
h = Hmm.loadHmm(os.path.join(ROOT, 'nanopore_amd', 'mappers', 'blasr_hmm_0.txt'))
n = int(sys.argv[1]); L = int(sys.argv[2]); W = int(sys.argv[3])
w = synth.make_workload(7, n, L, h.transitions, h.emissions, flank=0)
ctx = R.Context(0); ctx.set_hmm(h)
P = R.make_params(band_mode=1, fixed_width=W) if W > 0 else R.make_params(band_mode=0, max_pairs_per_base=24)  # W = 0: the reference's own band
b = ctx.stage_csr(P, w['ref'], w['ref_off'], w['read'], w['read_off'], w['guide_ops'], w['guide_off'])
st = b.stats()
print('class cells', [int(v) for v in b.class_stats()[1]], flush=True)
ms = [b.run() for _ in range(2)]
print('cells', st['cells'], 'ms', ms, flush=True)
