import sys, time, faulthandler
faulthandler.enable()
faulthandler.dump_traceback_later(15, exit=True)
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
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
