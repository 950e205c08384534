"""MarginAlignSnpCaller: SNP calling from posterior match probabilities marginalised over alignments
(nanopore/analyses/marginAlignSnpCaller.py:14-308).

The reference runs `cactus_realign --outputAllPosteriorProbs` once per read, hmm type, coverage and replicate
(:135-146) and adds every `refPos readPos prob` line into per-position base expectations (:149-155).  A read's
posteriors do not depend on which other reads are sampled, so here each hmm type costs ONE batched GPU call
(NPR_MODE_ALL_POSTERIORS, --splitMatrixBiggerThanThis=100) over all reads; the coverage / replicate loops only
re-sample which reads' posteriors are accumulated.  Output: marginaliseConsensus.xml with the reference's element
and attribute names (:268-303).
"""
import math
import os
import random
import xml.etree.ElementTree as ET
from itertools import product

import numpy as np

from .. import sam as pysam
from ..hmm import Hmm
from .abstractAnalysis import AbstractAnalysis
from .alignmentUncertainty import prettyXml
from .utils import (ANALYSIS_SPLIT_MATRIX_BIGGER_THAN, alignedPairs, getFastaDictionary, getFastqDictionary,
                    realignRecords, samIterator, trainedModelPath)

bases = "ACGT"
_BASE_CODE = np.full(256, -1, dtype=np.int64)
for _k, _c in enumerate(bases):
    _BASE_CODE[ord(_c)] = _BASE_CODE[ord(_c.lower())] = _k


# ---- substitution models: 4 x 4 arrays [from, to] over ACGT; the dict forms keep the reference's function names ----

def errorMatrixOfHmm(hmmFile):
    """P(observed base | true base) from a model's match emissions: the 4 x 4 block, every row scaled to sum 1
    (marginAlignSnpCaller.py:25-29)."""
    block = np.asarray(Hmm.loadHmm(hmmFile).emissions[:16], dtype=np.float64).reshape(4, 4)
    return block / block.sum(axis=1, keepdims=True)


NULL_MATRIX = np.ones((4, 4))                                      # no evolutionary prior (:31-32)
FLAT_ERROR_MATRIX = np.where(np.eye(4, dtype=bool), 0.8, 0.2 / 3)  # Jukes-Cantor-like (:34-35)


def _asDict(matrix):
    return {(bases[i], bases[j]): float(matrix[i, j]) for i in range(4) for j in range(4)}


def loadHmmErrorSubstitutionMatrix(hmmFile):
    return _asDict(errorMatrixOfHmm(hmmFile))


def getNullSubstitutionMatrix():
    return _asDict(NULL_MATRIX)


def getJukesCantorTypeSubstitutionMatrix():
    return _asDict(FLAT_ERROR_MATRIX)


def getProb(subMatrix, start, end):
    return subMatrix[(start, end)]


def basePosteriors(observed, refCodes, evolutionMatrix, errorMatrix):
    """Posterior of each candidate true base at many positions at once (:18-23, vectorised): observed[p, o] = fraction of
    the position's evidence that says base o, refCodes[p] = the reference base there.
    log P(candidate c) = log evolution[ref, c] + sum_o observed[o] * log error[c, o], normalised over c."""
    with np.errstate(divide="ignore"):
        logp = np.log(evolutionMatrix)[refCodes] + observed @ np.log(errorMatrix).T
    logp -= logp.max(axis=1, keepdims=True)
    w = np.exp(logp)
    return w / w.sum(axis=1, keepdims=True)


def calcBasePosteriorProbs(baseObservations, refBase, evolutionarySubstitionMatrix, errorSubstutionMatrix):
    """The one-position dict form of basePosteriors (the reference's signature)."""
    toArray = lambda d: np.array([[d[(a, b)] for b in bases] for a in bases], dtype=np.float64)  # noqa: E731
    post = basePosteriors(np.array([[baseObservations[b] for b in bases]], dtype=np.float64), np.array([bases.index(refBase)]),
                          toArray(evolutionarySubstitionMatrix), toArray(errorSubstutionMatrix))[0]
    return dict(zip(bases, post.tolist()))


class SnpCalls(object):
    """A set of SNP calls, each with the posterior probability it was made with, and its precision / recall curves over the
    101 probability thresholds 0.00 .. 1.00 (:163-197): a call counts at every threshold up to its (rounded) probability."""

    def __init__(self, totalHeldOut):
        self.totalHeldOut = totalHeldOut
        self.truePositives = []   # (probability, reference position)
        self.falsePositives = []
        self.falseNegatives = []
        self.notCalled = 0

    @staticmethod
    def bucket(calls):
        """[number of calls with round(100 p) >= t for t in 0..100], as floats."""
        p = np.asarray(list(calls), dtype=np.float64)
        hist = np.bincount(np.rint(p * 100).astype(np.int64), minlength=101).astype(np.float64) if len(p) else np.zeros(101)
        return hist[::-1].cumsum()[::-1].tolist()

    def addCalls(self, probabilities, positions, isTrue):
        for p, x, t in zip(probabilities.tolist(), positions.tolist(), isTrue.tolist()):
            (self.truePositives if t else self.falsePositives).append((p, x))

    def getPrecisionByProbability(self):
        tp = np.array(self.bucket(p for p, _ in self.truePositives))
        fp = np.array(self.bucket(p for p, _ in self.falsePositives))
        made = tp + fp
        return [float(t / m) if m else 0 for t, m in zip(tp, made)]

    def getRecallByProbability(self):
        return [t / self.totalHeldOut if self.totalHeldOut else 0 for t in self.bucket(p for p, _ in self.truePositives)]


def loadHeldOutSnps(referenceFastaFile, refSequences):
    """The truth behind a mutated reference: `<reference>_Index.txt` holds every sequence twice, as it was and (name +
    "_mutated") as the reference FASTA has it (written by mutate_reference.py; read at :60-78).  Returns
    {(sequence name, position): true base} for every position where the two differ; {} when there is no such file."""
    truthFile = referenceFastaFile + "_Index.txt"
    if not os.path.exists(truthFile):
        return {}
    pairs = getFastaDictionary(truthFile)
    originals = [n for n in pairs if n in refSequences]
    strays = [n for n in pairs if n not in refSequences and not n.endswith("_mutated")]
    assert not strays, "unexpected sequences in %s: %s" % (truthFile, strays)
    assert len(originals) == len(refSequences), "%s does not cover the reference" % truthFile
    held = {}
    for name in originals:
        true, mutated = pairs[name], pairs[name + "_mutated"]
        assert mutated == refSequences[name], "%s_mutated is not the reference sequence" % name
        a = np.frombuffer(true.encode(), dtype=np.uint8)
        b = np.frombuffer(mutated.encode(), dtype=np.uint8)
        for i in np.flatnonzero(a != b).tolist():
            held[(name, i)] = true[i]
    return held


class MarginAlignSnpCaller(AbstractAnalysis):
    hmmTypes = ("cactus", "trained_0", "trained_20", "trained_40")
    coverages = (1000000, 120, 60, 30, 10)

    def run(self, ctx=None, seed=None):
        from .. import realign
        AbstractAnalysis.run(self)
        rng = random.Random(seed)
        refSequences = getFastaDictionary(self.referenceFastaFile)
        readSequences = getFastqDictionary(self.readFastqFile)
        refNames = sorted(refSequences)
        refIndex = {n: i for i, n in enumerate(refNames)}
        totalReferenceLength = sum(len(s) for s in refSequences.values())
        snpSet = loadHeldOutSnps(self.referenceFastaFile, refSequences)
        hmmErrorMatrix = errorMatrixOfHmm(trainedModelPath("blasr_hmm_20.txt"))  # :56
        hmmFiles = {"cactus": None, "trained_0": trainedModelPath("blasr_hmm_0.txt"),
                    "trained_20": trainedModelPath("blasr_hmm_20.txt"), "trained_40": trainedModelPath("blasr_hmm_40.txt")}
        sam = pysam.Samfile(self.samFile, "r")
        records = list(samIterator(sam))
        for aR in records:  # :127-130
            refSeq = refSequences[sam.getrname(aR.rname)]
            assert aR.pos == 0 and aR.qstart == 0 and aR.qend == len(aR.query) and aR.aend == len(refSeq)
        # frequencies of aligned bases per read, from the SAM alignment itself (:111-118)
        alignedBases = []
        for aR in records:
            refName = sam.getrname(aR.rname)
            pos, code = [], []
            query = aR.query.upper()
            for r, q in alignedPairs(aR, len(refSequences[refName])):
                pos.append(r)
                code.append(bases.find(query[q]))
            alignedBases.append((refIndex[refName], np.array(pos, dtype=np.int64), np.array(code, dtype=np.int64)))
        node = ET.Element("marginAlignComparison")
        for hmmType in self.hmmTypes:
            # one batched GPU call per hmm type: all posterior match probabilities of every read (:135-146).  The pairs
            # stay in HBM: each coverage sample below scatter-adds them per reference position on the device
            # (npr_batch_base_expectations) instead of parsing a `refPos readPos prob` file per read (:149-155).
            from .utils import _context, _loadHmmInto, stageSamFile
            ctx = ctx or _context()
            _loadHmmInto(ctx, hmmFiles[hmmType])
            batch, stagedSam, _ = stageSamFile(self.samFile, self.referenceFastaFile, ANALYSIS_SPLIT_MATRIX_BIGGER_THAN,
                                               mode=realign.MODE_ALL_POSTERIORS, ctx=ctx, maxPairsPerBase=48)
            stagedSam.close()
            try:
                self._callsOfOneModel(node, hmmType, batch, records, refSequences, readSequences, refNames, alignedBases, snpSet,
                                      hmmErrorMatrix, totalReferenceLength, rng)
            finally:
                batch.close()
        sam.close()
        with open(os.path.join(self.outputDir, "marginaliseConsensus.xml"), "w") as fh:
            fh.write(prettyXml(node))
        self.finish()
        return node

    def _callsOfOneModel(self, node, hmmType, batch, records, refSequences, readSequences, refNames, alignedBases, snpSet, hmmErrorMatrix,
                         totalReferenceLength, rng):
        """All coverages and replicates for one hmm type, on the posteriors of one batched GPU pass."""
        batch.run()
        batch.finish()
        status = batch.results()["status"]
        for aR, st in zip(records, status):
            if st != 0:
                raise RuntimeError("Posterior computation failed for %s: status %d" % (aR.qname, st))
        refLengths = [len(refSequences[n]) for n in refNames]
        refRows = np.concatenate([[0], np.cumsum(refLengths)])
        for coverage in self.coverages:
            for replicate in range(3 if coverage < 1000000 else 1):
                order = list(range(len(records)))
                rng.shuffle(order)  # :91
                frequencies = [np.zeros((len(refSequences[n]), 4)) for n in refNames]
                seenF = [np.zeros(len(refSequences[n]), dtype=bool) for n in refNames]
                totalSampledReads = totalAlignedPairs = totalReadLength = 0
                sampled = np.zeros(len(records), dtype=np.uint8)
                for i in order:
                    if totalReadLength / totalReferenceLength >= coverage:  # :94
                        break
                    aR = records[i]
                    totalReadLength += len(readSequences[aR.qname])
                    totalSampledReads += 1
                    k, pos, code = alignedBases[i]
                    totalAlignedPairs += len(pos)
                    seenF[k][pos] = True
                    ok = code >= 0
                    np.add.at(frequencies[k], (pos[ok], code[ok]), 1.0)
                    sampled[i] = 1
                allE, allSeen = batch.base_expectations(refLengths, use=sampled)
                expectations = [allE[refRows[k]:refRows[k + 1]] for k in range(len(refNames))]
                seenE = [allSeen[refRows[k]:refRows[k + 1]] for k in range(len(refNames))]
                totalHeldOut = len(snpSet)
                totalNotHeldOut = totalReferenceLength - totalHeldOut
                callSets = [SnpCalls(totalHeldOut) for _ in range(4)]
                configs = ((FLAT_ERROR_MATRIX, expectations, seenE), (hmmErrorMatrix, expectations, seenE),
                           (FLAT_ERROR_MATRIX, frequencies, seenF), (hmmErrorMatrix, frequencies, seenF))
                for k, refSeqName in enumerate(refNames):
                    # every position of the sequence at once: candidate posteriors, then one call per non-reference base
                    refCodes = _BASE_CODE[np.frombuffer(refSequences[refSeqName].encode(), dtype=np.uint8)]
                    trueCodes = refCodes.copy()
                    for (name, i), base in snpSet.items():
                        if name == refSeqName:
                            trueCodes[i] = _BASE_CODE[ord(base)]
                    for (errorM, table, seen), snpCalls in zip(configs, callSets):
                        snpCalls.notCalled += int((~seen[k]).sum())
                        total = table[k].sum(axis=1)
                        where = np.flatnonzero(seen[k] & (total > 0.0) & (refCodes >= 0))
                        if not len(where):
                            continue
                        post = basePosteriors(table[k][where] / total[where, None], refCodes[where], NULL_MATRIX, errorM)
                        for c in range(4):
                            called = refCodes[where] != c
                            isTrue = (trueCodes[where] != refCodes[where]) & (trueCodes[where] == c)
                            snpCalls.addCalls(post[called, c], where[called], isTrue[called])
                tags = ("marginAlignMaxExpectedSnpCalls", "marginAlignMaxLikelihoodSnpCalls", "maxFrequencySnpCalls",
                        "maximumLikelihoodSnpCalls")
                for snpCalls, tagName in zip(callSets, tags):
                    recall = snpCalls.getRecallByProbability()
                    precision = snpCalls.getPrecisionByProbability()
                    fScore, pIndex = max((2 * recall[i] * precision[i] / (recall[i] + precision[i])
                                          if recall[i] + precision[i] > 0 else 0.0, i) for i in range(len(recall)))
                    ET.SubElement(node, tagName + "_" + hmmType, {
                        "coverage": str(coverage), "actualCoverage": str(float(totalAlignedPairs) / totalReferenceLength),
                        "totalAlignedPairs": str(totalAlignedPairs), "totalReferenceLength": str(totalReferenceLength),
                        "replicate": str(replicate), "totalReads": str(len(records)),
                        "avgSampledReadLength": str(float(totalReadLength) / max(totalSampledReads, 1)),
                        "totalSampledReads": str(totalSampledReads), "totalHeldOut": str(totalHeldOut),
                        "totalNonHeldOut": str(totalNotHeldOut), "recall": str(recall[pIndex]),
                        "precision": str(precision[pIndex]), "fScore": str(fScore),
                        "optimumProbThreshold": str(float(pIndex) / 100.0), "totalNoCalls": str(snpCalls.notCalled),
                        "recallByProbability": " ".join(map(str, recall)),
                        "precisionByProbability": " ".join(map(str, precision))})
