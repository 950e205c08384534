import sys, time, faulthandler
faulthandler.enable()
faulthandler.dump_traceback_later(15, exit=True)
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from helpers import *
import os
from nanopore_amd import _lib
if os.environ.get("NPR_LIB"): _lib.LIB_PATH = os.environ["NPR_LIB"]
from nanopore_amd import realign as R
from nanopore_amd.hmm import Hmm
t0 = time.time()
print('create ctx', flush=True)
ctx = R.Context(0)
print('ctx ok', time.time()-t0, flush=True)
ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + '/blasr_hmm_0.txt'))
print('hmm ok', flush=True)
rng = np.random.default_rng(1)
X, Y, ops = random_pair(rng, 50)
P = R.make_params(band_mode=1, fixed_width=20)
b = ctx.stage(P, [bytes(b"ACGT"[c] for c in X)], [bytes(b"ACGT"[c] for c in Y)], [ops])
print('staged', b.stats(), flush=True)
ms = b.run()
print('ran', ms, flush=True)
b.finish()
print('finished', b.results(), flush=True)
print(b.ops())
off, x, y, p = b.pairs()
print('pairs', len(x), 'lX', len(X), 'lY', len(Y))
print(np.stack([x, y]).T[:30].tolist())
print(p[:30].tolist())
os.environ['NPR_KERNEL'] = 'generic'
b2 = ctx.stage(P, [bytes(b"ACGT"[c] for c in X)], [bytes(b"ACGT"[c] for c in Y)], [ops])
b2.run(); b2.finish(); off2, x2, y2, p2 = b2.pairs()
print('generic pairs', len(x2)); print(np.stack([x2, y2]).T[:30].tolist()); print(p2[:30].tolist())
