"""Expectation maximisation of the five-state pair-HMM.

Stands in for cactus.bar.cactus_expectationMaximisation, which the reference drives from
nanopore/analyses/utils.py:471-531 (`learnModelFromSamFileTargetFn`) and which is absent from the snapshot.  Its
option names are kept (`Options`, utils.py:509-523).  Where the original runs `cactus_realign --outputExpectations`
over chunks of alignments as jobTree jobs in every iteration, here one staged GPU batch serves every iteration of
every trial: `Batch.expectations()` is the E-step (C ABI `npr_batch_expectations`), the M-step is the few lines of
`normalise` below.  Outputs: the two-line model file and the `hmm.txt.xml` that nanopore/analyses/hmm.py:15-86 and
nanopore/metaAnalyses/hmmMetaAnalysis.py read (`transition from/to/avg/std`, `emission state/x/y/avg/std`,
`hmm runningLikelihoods`).
"""
import xml.etree.ElementTree as ET
from xml.dom import minidom

import numpy as np

from .hmm import Hmm, STATE_NUMBER, SYMBOL_NUMBER

# the transitions of the five-state cell update (from, to); everything else stays 0
USED_TRANSITIONS = [(0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (1, 0), (1, 1), (1, 2), (2, 0), (2, 1), (2, 2), (3, 0), (3, 3),
                    (4, 0), (4, 4)]
# the shipped models carry no short-gap switches (13 transitions); EM from a random start trains all 15


class Options(object):
    """Option names of cactus_expectationMaximisation.Options as set at utils.py:509-523."""

    def __init__(self):
        self.modelType = "fiveStateAsymmetric"
        self.optionsToRealign = "--diagonalExpansion=10 --splitMatrixBiggerThanThis=300"
        self.randomStart = True
        self.trials = 3
        self.outputTrialHmms = True
        self.iterations = 100
        self.maxAlignmentLengthPerJob = 700000       # alignment columns of one jobTree job of the reference's trainer (utils.py:516)
        self.maxAlignmentLengthToSample = 50000000   # alignment columns the trainer samples from the SAM (utils.py:517)
        self.jobsPerBatch = 64  # this build: such jobs staged as ONE GPU batch (the reference runs them as separate processes)
        self.outputXMLModelFile = None
        self.trainEmissions = True
        self.seed = None  # this build: seed of the random starts (None = nondeterministic, like the reference)


def sampleAlignments(lengths, options, rng=None):
    """Which alignments the trainer looks at and how they are cut into batches (options.maxAlignmentLengthToSample /
    maxAlignmentLengthPerJob, utils.py:516-517): the alignments are taken in random order until the next one would take the sum
    of their lengths (alignment columns) past maxAlignmentLengthToSample -- at least one is always taken --, then cut into batches
    of at most jobsPerBatch x maxAlignmentLengthPerJob columns (an alignment longer than that is a batch of its own).  Returns a
    list of index arrays (each sorted: file order inside a batch).  A SAM below the limits comes back whole, as one batch."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    if n == 0:
        return []
    limit = int(options.maxAlignmentLengthToSample)
    if lengths.sum() <= limit:
        taken = np.arange(n)
    else:
        rng = rng if rng is not None else np.random.default_rng(options.seed)
        order = rng.permutation(n)
        k = int(np.searchsorted(np.cumsum(lengths[order]), limit, side="right"))
        taken = order[:max(k, 1)]
    per_batch = max(1, int(options.maxAlignmentLengthPerJob)) * max(1, int(getattr(options, "jobsPerBatch", 1)))
    batches, cur, cur_len = [], [], 0
    for i in taken:
        li = int(lengths[i])
        if cur and cur_len + li > per_batch:
            batches.append(np.sort(np.array(cur, dtype=np.int64)))
            cur, cur_len = [], 0
        cur.append(int(i))
        cur_len += li
    if cur:
        batches.append(np.sort(np.array(cur, dtype=np.int64)))
    return batches


class BatchSet(object):
    """Several staged batches that the E-step treats as one: expected counts and log-likelihoods are summed (what the
    reference's trainer does with the --outputExpectations files of its jobs)."""

    def __init__(self, batches):
        self.batches = list(batches)
        self.ctx = self.batches[0].ctx

    def expectations(self):
        T = E = ll = None
        ms = 0.0
        for b in self.batches:
            t, e, l, m = b.expectations()
            T, E, ll = (t, e, l) if T is None else (T + t, E + e, ll + l)
            ms += m
        return T, E, ll, ms

    def stats(self):
        out = {}
        for b in self.batches:
            for k, v in b.stats().items():
                out[k] = out.get(k, 0) + v
        return out

    def close(self):
        for b in self.batches:
            b.close()


def randomise(hmm, rng):
    """Random start: random probabilities on the used transitions, random emissions, everything normalised."""
    T = np.zeros((STATE_NUMBER, STATE_NUMBER))
    for a, b in USED_TRANSITIONS:
        T[a, b] = rng.random() + 1e-3
    T /= T.sum(axis=1, keepdims=True)
    hmm.transitions = [float(v) for v in T.reshape(-1)]
    k = SYMBOL_NUMBER * SYMBOL_NUMBER
    E = rng.random((STATE_NUMBER, k)) + 1e-3
    E /= E.sum(axis=1, keepdims=True)
    hmm.emissions = [float(v) for v in E.reshape(-1)]
    return hmm


def normalise(hmm, T_exp, E_exp, trainEmissions=True):
    """M-step: transition rows and (optionally) each state's emission block renormalised from expected counts."""
    T = np.array(T_exp, dtype=np.float64).reshape(STATE_NUMBER, STATE_NUMBER)
    old = np.array(hmm.transitions).reshape(STATE_NUMBER, STATE_NUMBER)
    for a in range(STATE_NUMBER):
        s = T[a].sum()
        old[a] = T[a] / s if s > 0 else old[a]
    hmm.transitions = [float(v) for v in old.reshape(-1)]
    if trainEmissions:
        k = SYMBOL_NUMBER * SYMBOL_NUMBER
        E = np.array(E_exp, dtype=np.float64).reshape(STATE_NUMBER, k)
        cur = np.array(hmm.emissions).reshape(STATE_NUMBER, k)
        for s_ in range(STATE_NUMBER):
            tot = E[s_].sum()
            if tot > 0:
                cur[s_] = E[s_] / tot
        hmm.emissions = [float(v) for v in cur.reshape(-1)]
    return hmm


def _collectiveDevice(device, group):
    """Where a collective's tensor has to live: the caller's choice, else the rank's GPU under backend nccl (= RCCL, which
    takes device tensors only), else the host."""
    import torch
    import torch.distributed as dist
    if device is not None:
        return torch.device(device)
    if str(dist.get_backend(group)) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _isRankZero(group=None):
    try:
        import torch.distributed as dist
    except ImportError:
        return True
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank(group) == 0


def _barrier(group=None):
    try:
        import torch.distributed as dist
    except ImportError:
        return
    if dist.is_available() and dist.is_initialized():
        dist.barrier(group=group)


def allReduceExpectations(T, E, ll, device=None, group=None):
    """Sharded EM (one process per GPU, each with its own reads): the per-rank expected counts and log-likelihoods
    are summed over the ranks -- the one collective the training loop needs, 25 + 80 + 1 doubles per model slot per
    iteration (RCCL over xGMI under backend `nccl`).  A no-op outside torch.distributed."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:
        return T, E, ll
    if not (dist.is_available() and dist.is_initialized()):
        return T, E, ll
    flat = np.concatenate([np.asarray(T).reshape(-1), np.asarray(E).reshape(-1), np.asarray(ll).reshape(-1)])
    t = torch.from_numpy(flat.copy()).to(_collectiveDevice(device, group))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    flat = t.cpu().numpy()
    nT, nE = np.asarray(T).size, np.asarray(E).size
    return (flat[:nT].reshape(np.asarray(T).shape), flat[nT:nT + nE].reshape(np.asarray(E).shape),
            flat[nT + nE:].reshape(np.asarray(ll).shape))


def broadcastModel(hmm, device=None, group=None, src=0):
    """Sharded EM: every rank continues from rank `src`'s model (a random start drawn per rank would make the first
    E-step sum counts computed under different models).  A no-op outside torch.distributed."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:
        return hmm
    if not (dist.is_available() and dist.is_initialized()):
        return hmm
    t = torch.tensor(list(hmm.transitions) + list(hmm.emissions) + [float(hmm.likelihood)], dtype=torch.float64).to(_collectiveDevice(device, group))
    dist.broadcast(t, src=src, group=group)
    v = t.cpu().numpy()
    nT = len(hmm.transitions)
    hmm.transitions = [float(x) for x in v[:nT]]
    hmm.emissions = [float(x) for x in v[nT:-1]]
    hmm.likelihood = float(v[-1])
    return hmm


def expectationMaximisation(batch, hmm, iterations, trainEmissions=True, slot=0, log=None, reduce_device=None):
    """Runs `iterations` EM iterations on a staged batch whose reads all use model `slot`; updates `hmm` in place.
    Returns the running likelihoods (log-likelihood of the data under the model BEFORE each update).  Under
    torch.distributed every rank holds a shard of the reads and the expectations are all-reduced, so all ranks
    walk through identical models."""
    running = []
    for it in range(iterations):
        batch.ctx.set_hmm(hmm, slot=slot)
        T, E, ll, _ = batch.expectations()
        T, E, ll = allReduceExpectations(T, E, ll, device=reduce_device)
        hmm.likelihood = float(ll[slot])
        running.append(hmm.likelihood)
        normalise(hmm, T[slot], E[slot], trainEmissions)
        if log:
            log("EM iteration %d: log-likelihood %s" % (it, hmm.likelihood))
    return running


def writeXML(path, trialHmms, runningLikelihoods):
    """hmm.txt.xml: per-transition / per-emission mean and standard deviation over trials plus every trial's
    running likelihoods (schema read by nanopore/analyses/hmm.py:31-36, :43-44, :62-66, :80-82)."""
    root = ET.Element("hmms")
    Ts = np.array([h.transitions for h in trialHmms])
    Es = np.array([h.emissions for h in trialHmms])
    for a in range(STATE_NUMBER):
        for b in range(STATE_NUMBER):
            v = Ts[:, a * STATE_NUMBER + b]
            ET.SubElement(root, "transition", {"from": str(a), "to": str(b), "avg": str(float(v.mean())), "std": str(float(v.std()))})
    bases = "ACGT"
    k = SYMBOL_NUMBER * SYMBOL_NUMBER
    for s in range(STATE_NUMBER):
        for x in range(SYMBOL_NUMBER):
            for y in range(SYMBOL_NUMBER):
                v = Es[:, s * k + x * SYMBOL_NUMBER + y]
                ET.SubElement(root, "emission", {"state": str(s), "x": bases[x], "y": bases[y], "avg": str(float(v.mean())),
                                                 "std": str(float(v.std()))})
    for h, rl in zip(trialHmms, runningLikelihoods):
        ET.SubElement(root, "hmm", {"runningLikelihoods": " ".join(str(v) for v in rl), "likelihood": str(h.likelihood)})
    with open(path, "w") as fh:
        fh.write(minidom.parseString(ET.tostring(root, "utf-8")).toprettyxml(indent="  "))


def expectationMaximisationTrials(batch, outputModel, options, startHmm=None, log=None, reduce_device=None):
    """`options.trials` independent EM runs (random starts when options.randomStart), the trial with the highest
    final likelihood is written to `outputModel`; the XML summary goes to options.outputXMLModelFile.  Under
    torch.distributed (every rank a shard of the reads) all ranks walk through the same models and pick the same trial:
    each trial starts from rank 0's model and every likelihood is the sum over the ranks; the files (model, trial models,
    XML) are written by rank 0 alone, and every rank returns once they exist."""
    rng = np.random.default_rng(options.seed)
    trialHmms, running = [], []
    for trial in range(options.trials):
        hmm = Hmm() if startHmm is None else startHmm.copy()
        if options.randomStart or startHmm is None:
            randomise(hmm, rng)
        broadcastModel(hmm, device=reduce_device)
        rl = expectationMaximisation(batch, hmm, options.iterations, options.trainEmissions, log=log, reduce_device=reduce_device)
        # likelihood of the final parameters
        batch.ctx.set_hmm(hmm)
        T, E, ll, _ = batch.expectations()
        _, _, ll = allReduceExpectations(T, E, ll, device=reduce_device)
        hmm.likelihood = float(ll[0])
        rl.append(hmm.likelihood)
        trialHmms.append(hmm)
        running.append(rl)
        if options.outputTrialHmms and _isRankZero():
            hmm.write("%s_%d" % (outputModel, trial))
    best = max(trialHmms, key=lambda h: h.likelihood)
    if _isRankZero():  # one writer: the ranks hold identical models, and a shared filesystem has one file per path
        best.write(outputModel)
        if options.outputXMLModelFile:
            writeXML(options.outputXMLModelFile, trialHmms, running)
    _barrier()
    return best, trialHmms, running
