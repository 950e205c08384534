"""Randomised GPU-vs-oracle parity stress (bring-up; the committed tests run a fixed subset of this):
python tools/gpu_parity_stress.py [seconds] [seed].  Every read: bit-exact against the fp32 mirror, 1e-4 against fp64."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from nanopore_amd import realign as R
import test_gpu_parity as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = R.Context(0)
t0 = time.time(); cases = reads = skipped = 0
classes = np.zeros(16, dtype=np.int64)
while time.time() - t0 < budget:
    kind = int(rng.integers(0, 4))
    if kind == 0:    # narrow fixed bands, short reads, heavy indels: rebases in both directions
        kw = dict(band_mode=1, fixed_width=int(rng.integers(2, 130)))
        args = (int(rng.integers(2, 12)), 1, int(rng.integers(5, 700)))
        extra = dict(indel=float(rng.random() * 0.35), max_indel=int(rng.integers(1, 80)))
    elif kind == 1:  # R = 2 / 4 bands
        kw = dict(band_mode=1, fixed_width=int(rng.integers(130, 500)))
        args = (int(rng.integers(2, 6)), 50, int(rng.integers(200, 1500)))
        extra = dict(indel=float(rng.random() * 0.3), max_indel=int(rng.integers(1, 120)))
    elif kind == 2:  # wide fixed bands: k_dp_wide
        kw = dict(band_mode=1, fixed_width=int(rng.integers(520, 2400)))
        args = (int(rng.integers(1, 4)), 400, int(rng.integers(600, 1800)))
        extra = dict(indel=float(rng.random() * 0.3), max_indel=int(rng.integers(1, 200)))
    else:            # the reference's anchor band with random parameters
        kw = dict(band_mode=0, diagonal_expansion=int(rng.integers(0, 10)) * 2, constraint_trim=int(rng.integers(0, 20)),
                  split_threshold=int(rng.integers(10, 3000)), max_pairs_per_base=60)
        args = (int(rng.integers(2, 8)), 20, int(rng.integers(100, 2500)))
        extra = dict(indel=float(rng.random() * 0.35), max_indel=int(rng.integers(1, 150)))
    try:
        T._run_case(ctx, rng, args[0], args[1], args[2], kw, **extra)
    except AssertionError as e:
        # (the ORACLE's pair list has a fixed capacity; the product runs an overflowed read again with four times the room.  A read of 990 bases
        # against 2653 under a band of 1961 cells is such a case: statuses (0, -3, -3): nothing to compare)
        if e.args and isinstance(e.args[0], tuple) and e.args[0][:3] == (0, -3, -3):
            skipped += 1
            continue
        print("FAILED case", cases, "seed", seed, kw, args, extra, flush=True)
        raise
    except Exception:
        print("FAILED case", cases, "seed", seed, kw, args, extra, flush=True)
        raise
    cases += 1; reads += args[0]
print("parity stress ok: %d cases, %d reads, %.0f s (%d cases skipped: the oracle's pair list overflowed)" % (cases, reads, time.time() - t0, skipped))
