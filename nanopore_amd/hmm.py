"""Five-state pair-HMM parameter type and file format.

Stands in for `Hmm` / `SYMBOL_NUMBER` of cactus.bar.cactus_expectationMaximisation, which the reference
imports at nanopore/analyses/utils.py:3-4, scripts/modifyHmm.py:2 and
nanopore/analyses/marginAlignSnpCaller.py:11 but which is absent from the snapshot (empty
submodules/cactus).  Attribute surface relied on by the reference: `stateNumber`, `emissions`
(flat list of 80), `transitions` (flat list of 25), `likelihood`, `Hmm.loadHmm(path)`,
`hmm.write(path)`.

File layout (pinned by nanopore/mappers/blasr_hmm_{0,20,40}.txt):
  line 1: <type> <25 transition probabilities, row-major T[from*5+to]> <likelihood>
  line 2: <80 emission probabilities E[state*16 + x*4 + y]>, x = reference base, y = read base, ACGT
States: 0 match, 1 shortGapX (ref only), 2 shortGapY (read only), 3 longGapX, 4 longGapY
(nanopore/analyses/utils.py:617, nanopore/analyses/hmm.py:24-28).
"""
SYMBOL_NUMBER = 4
STATE_NUMBER = 5

FIVE_STATE_ASYMMETRIC = 1  # the `type` token of the shipped model files

MATCH, SHORT_GAP_X, SHORT_GAP_Y, LONG_GAP_X, LONG_GAP_Y = range(5)


def _fmt(v):
    """Python-2 ``str(float)``: 12 significant digits, always with a '.0' or an exponent, and exponent
    notation from 1e11 upwards -- the notation of the shipped model files (e.g. the likelihood field
    ``-6.14898969935e+11`` of nanopore/mappers/blasr_hmm_0.txt:1)."""
    s = "%.12g" % v
    if "e" not in s and "n" not in s and abs(v) >= 1e11:
        mant, exp = ("%.11e" % v).split("e")
        mant = mant.rstrip("0").rstrip(".")
        s = "%se%s%02d" % (mant, exp[0], int(exp[1:]))
    if "." not in s and "e" not in s and "n" not in s:
        s += ".0"
    return s


class Hmm(object):
    def __init__(self, modelType=FIVE_STATE_ASYMMETRIC):
        self.type = int(modelType)
        self.stateNumber = STATE_NUMBER
        self.transitions = [0.0] * (STATE_NUMBER * STATE_NUMBER)
        self.emissions = [0.0] * (STATE_NUMBER * SYMBOL_NUMBER * SYMBOL_NUMBER)
        self.likelihood = 0.0

    @staticmethod
    def loadHmm(path):
        with open(path) as fh:
            lines = [ln.split() for ln in fh if ln.strip()]
        if len(lines) < 2:
            raise RuntimeError("Malformed HMM file %s: expected two lines" % path)
        head, emis = lines[0], lines[1]
        if len(head) != 2 + STATE_NUMBER * STATE_NUMBER:
            raise RuntimeError("Malformed HMM file %s: %d tokens on the transition line" % (path, len(head)))
        if len(emis) != STATE_NUMBER * SYMBOL_NUMBER * SYMBOL_NUMBER:
            raise RuntimeError("Malformed HMM file %s: %d emissions" % (path, len(emis)))
        hmm = Hmm(int(head[0]))
        hmm.transitions = [float(t) for t in head[1:-1]]
        hmm.likelihood = float(head[-1])
        hmm.emissions = [float(e) for e in emis]
        return hmm

    def write(self, path):
        with open(path, "w") as fh:
            fh.write(" ".join([str(self.type)] + [_fmt(t) for t in self.transitions] + [_fmt(self.likelihood)]))
            fh.write("\n")
            fh.write(" ".join(_fmt(e) for e in self.emissions))
            fh.write("\n")

    # convenience views -----------------------------------------------------------------------
    def transitionMatrix(self):
        n = self.stateNumber
        return [self.transitions[i * n:(i + 1) * n] for i in range(n)]

    def emissionMatrix(self, state):
        k = SYMBOL_NUMBER * SYMBOL_NUMBER
        block = self.emissions[state * k:(state + 1) * k]
        return [block[i * SYMBOL_NUMBER:(i + 1) * SYMBOL_NUMBER] for i in range(SYMBOL_NUMBER)]

    def copy(self):
        h = Hmm(self.type)
        h.transitions = list(self.transitions)
        h.emissions = list(self.emissions)
        h.likelihood = self.likelihood
        return h


def stockHmm():
    """The model used when no --loadHmm is given (hmmFile=None, nanopore/mappers/abstractMapper.py:36-37).

    Its values live in the absent cactus C source, so they are UNPINNED (SURVEY.md 8c "Stock model");
    these are the recalled cPecan defaults, documented in DESIGN.md.
    """
    h = Hmm()
    cont, so, se, sw, lo_, le = 0.9703833696510062, 0.0129868352330243, 0.7126062401851738, \
        0.0073673675173412815, (1.0 - 0.9703833696510062 - 2 * 0.0129868352330243) / 2.0, 0.99656342579062
    T = [[0.0] * 5 for _ in range(5)]
    T[0] = [cont, so, so, lo_, lo_]
    T[1] = [1.0 - se - sw, se, sw, 0.0, 0.0]
    T[2] = [1.0 - se - sw, sw, se, 0.0, 0.0]
    T[3] = [1.0 - le, 0.0, 0.0, le, 0.0]
    T[4] = [1.0 - le, 0.0, 0.0, 0.0, le]
    h.transitions = [v for row in T for v in row]
    m, ts, tv = 0.12064298095701059, 0.018577373224845586, 0.010396478746046977
    kinds = {(0, 2), (2, 0), (1, 3), (3, 1)}  # transitions A<->G, C<->T
    em = []
    for x in range(4):
        for y in range(4):
            em.append(m if x == y else (ts if (x, y) in kinds else tv))
    tot = sum(em)
    h.emissions[0:16] = [e / tot for e in em]
    for s in range(1, 5):
        h.emissions[16 * s:16 * (s + 1)] = [1.0 / 16.0] * 16
    return h
