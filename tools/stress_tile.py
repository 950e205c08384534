"""Stress of the stripe kernel's hand-over between wavefronts: the same batch N times with freshly poisoned scratch
(NPR_POISON), pairs compared with the one-wavefront-per-task run.  Bring-up tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
from nanopore_amd import _lib, realign as R, synth
from nanopore_amd.hmm import Hmm
from helpers import load_model_arrays, MODEL_DIR
T, E, _ = load_model_arrays()
w = synth.make_workload(1007, 48, 3000, T, E, flank=0, length_sigma=0.5, len_min=300, len_max=9000)
ctx = R.Context(0); ctx.set_hmm(Hmm.loadHmm(MODEL_DIR + "/blasr_hmm_0.txt"))
P = R.make_params(band_mode=R.BAND_ANCHOR, diagonal_expansion=10, constraint_trim=14, split_threshold=3000)
def run():
    b = ctx.stage_csr(P, w["ref"], w["ref_off"], w["read"], w["read_off"], w["guide_ops"], w["guide_off"])
    ms = b.run(); b.finish()
    r = b.results(); pr = b.pairs()
    b.close()
    return r, pr, ms
ctx.set_option(_lib.OPTIONS['tile_waves'], 1)  # NPR_OPT_TILE_WAVES: one wavefront per task
good = run()
ctx.set_option(_lib.OPTIONS['tile_waves'], 0)
bad_ll = bad_pairs = 0
t0 = time.time(); tms = []
for rep in range(N):
    r, pr, ms = run(); tms.append(ms)
    if not (np.array_equal(r['loglik'], good[0]['loglik']) and np.array_equal(r['loglik_bwd'], good[0]['loglik_bwd'])): bad_ll += 1
    elif not all(np.array_equal(a, c) for a, c in zip(pr, good[1])):
        bad_pairs += 1
        poff, px, py, pp = pr; gpoff, gpx, gpy, gpp = good[1]
        for i in range(48):
            A = dict(zip(zip(px[poff[i]:poff[i+1]].tolist(), py[poff[i]:poff[i+1]].tolist()), pp[poff[i]:poff[i+1]].tolist()))
            G = dict(zip(zip(gpx[gpoff[i]:gpoff[i+1]].tolist(), gpy[gpoff[i]:gpoff[i+1]].tolist()), gpp[gpoff[i]:gpoff[i+1]].tolist()))
            if A != G:
                ex = sorted(set(A) - set(G)); mi = sorted(set(G) - set(A)); df = sorted(k for k in A if k in G and A[k] != G[k])
                lX = int(w['ref_off'][i+1]-w['ref_off'][i]); lY = int(w['read_off'][i+1]-w['read_off'][i])
                print('rep', rep, 'read', i, 'lX', lX, 'lY', lY, 'n', len(A), len(G), 'extra', len(ex), 'missing', len(mi), 'differ', len(df))
                for name, ks in (('extra', ex), ('missing', mi), ('differ', df)):
                    if ks:
                        a = np.array(ks); d = a[:,0] + a[:,1] + 2
                        print('   ', name, 'x+1 range', a[:,0].min()+1, a[:,0].max()+1, 'x+1 mod 128', sorted(set(((a[:,0]+1) % 128).tolist()))[:40], 'stripes', sorted(set(((a[:,0]+1)//128).tolist())), 'd', d.min(), d.max(), 'n_d', len(set(d.tolist())), [ (k, A.get(k), G.get(k)) for k in ks[:4]])
print('runs', N, 'bad totals', bad_ll, 'bad pairs only', bad_pairs, 'kernel ms median %.3f' % float(np.median(tms)), 'wall %.1f s' % (time.time() - t0), flush=True)
