"""How far could the reference's own arithmetic move a result?  cactus_realign (absent: parity unpinned) adds log-probabilities
with a piecewise-cubic approximation [RECALLED, SURVEY.md Appendix A]; this build's oracle is exact.  Runs the fp64 oracle
both ways on BASELINE.json configs[1] (all 1000 reads) and on a north-star sample, and reports the cigars that differ and the
largest posterior / score / log-likelihood differences -- the yardstick next to which "GPU vs fp64 oracle" (tests/
test_gpu_configs.py) should be read.  CPU only:  python tools/logadd_risk.py [n_northstar_reads]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_model_arrays, orc  # noqa: E402
from nanopore_amd import synth  # noqa: E402
from nanopore_amd.realign import encode  # noqa: E402


def compare(name, w, W, n):
    T, E, _ = load_model_arrays()
    h = orc.make_hmm(T, E)
    P = orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=W)
    if w.get("guide_start") is not None:
        lead, ilen = w["guide_start"][:n, 0], w["interval_len"][:n]
        X = encode(b"".join(bytes(w["ref"][w["ref_off"][i] + lead[i]:w["ref_off"][i] + lead[i] + ilen[i]]) for i in range(n)))
        x_off = np.concatenate([[0], np.cumsum(ilen)]).astype(np.int64)
    else:
        X, x_off = encode(bytes(w["ref"][:w["ref_off"][n]])), w["ref_off"][:n + 1]
    Y = encode(bytes(w["read"][:w["read_off"][n]]))
    args = (h, P, X, x_off, Y, w["read_off"][:n + 1], w["guide_ops"][:w["guide_off"][n]], w["guide_off"][:n + 1])
    exact = orc.realign_batch(*args, precision=0, threads=os.cpu_count(), native=True)
    with orc.logadd_kind(orc.LOGADD_APPROX):
        approx = orc.realign_batch(*args, precision=0, threads=os.cpu_count(), native=True)
    mirror = orc.realign_batch(*args, precision=1, threads=os.cpu_count(), native=True)
    same_a = sum(int(np.array_equal(a, b)) for a, b in zip(exact["ops"], approx["ops"]))
    same_m = sum(int(np.array_equal(a, b)) for a, b in zip(exact["ops"], mirror["ops"]))
    print("%s (%d reads): cigars identical exact-vs-approximate log-add %d/%d; exact fp64 vs fp32 mirror (= the GPU) %d/%d"
          % (name, n, same_a, n, same_m, n))
    print("    mean-posterior score: max |exact - approx| %.3g, max |exact - mirror| %.3g"
          % (np.abs(exact["score"] - approx["score"]).max(), np.abs(exact["score"] - mirror["score"]).max()))
    rel = lambda a, b: np.abs((a - b) / a).max()  # noqa: E731
    print("    log-likelihood: max rel. |exact - approx| %.3g, |exact - mirror| %.3g" % (rel(exact["total_ll"], approx["total_ll"]), rel(exact["total_ll"], mirror["total_ll"])))


if __name__ == "__main__":
    T, E, _ = load_model_arrays()
    w, W = synth.config_c2(T, E, n_reads=1000)
    compare("configs[1]: 1 kb reads, band 100", w, W, 1000)
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    w, W = synth.config_north_star(T, E, n_reads=k)
    compare("north star: 10 kb reads, band 200", w, W, k)
