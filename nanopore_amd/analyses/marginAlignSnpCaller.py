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


def getProb(subMatrix, start, end):
    return subMatrix[(start, end)]


def calcBasePosteriorProbs(baseObservations, refBase, evolutionarySubstitionMatrix, errorSubstutionMatrix):
    """Posterior of each candidate true base given the (fractional) observed base counts (:18-23)."""
    logBaseProbs = [math.log(getProb(evolutionarySubstitionMatrix, refBase, missingBase))
                    + sum(math.log(getProb(errorSubstutionMatrix, missingBase, observedBase)) * baseObservations[observedBase]
                          for observedBase in bases) for missingBase in bases]
    totalLogProb = logBaseProbs[0]
    for y in logBaseProbs[1:]:
        totalLogProb = totalLogProb + math.log(1 + math.exp(y - totalLogProb))
    return dict(zip(bases, [math.exp(lp - totalLogProb) for lp in logBaseProbs]))


def loadHmmErrorSubstitutionMatrix(hmmFile):
    """Match emissions of a model, each reference-base row normalised to 1 (:25-29)."""
    hmm = Hmm.loadHmm(hmmFile)
    m = hmm.emissions[:len(bases) ** 2]
    m = [m[i] / sum(m[4 * (i // 4):4 * (1 + i // 4)]) for i in range(len(m))]
    return dict(zip(product(bases, bases), m))


def getNullSubstitutionMatrix():
    return dict(zip(product(bases, bases), [1.0] * len(bases) ** 2))


def getJukesCantorTypeSubstitutionMatrix():
    return dict(zip(product(bases, bases), [0.8 if x[0] == x[1] else (0.2 / 3) for x in product(bases, bases)]))


class SnpCalls(object):
    """Call set with cumulative precision / recall by probability threshold (:163-197)."""

    def __init__(self, totalHeldOut):
        self.falsePositives = []
        self.truePositives = []
        self.falseNegatives = []
        self.notCalled = 0
        self.totalHeldOut = totalHeldOut

    @staticmethod
    def bucket(calls):
        buckets = [0.0] * 101
        for prob in calls:
            buckets[int(round(prob * 100))] += 1
        for i in range(len(buckets) - 2, -1, -1):
            buckets[i] += buckets[i + 1]
        return buckets

    def getPrecisionByProbability(self):
        tPs = self.bucket([x[0] for x in self.truePositives])
        fPs = self.bucket([x[0] for x in self.falsePositives])
        return [float(tPs[i]) / (tPs[i] + fPs[i]) if tPs[i] + fPs[i] != 0 else 0 for i in range(len(tPs))]

    def getRecallByProbability(self):
        return [i / self.totalHeldOut if self.totalHeldOut != 0 else 0 for i in self.bucket([x[0] for x in self.truePositives])]


def loadHeldOutSnps(referenceFastaFile, refSequences):
    """The `<reference>_Index.txt` truth file written by mutate_reference.py (:60-78)."""
    snpSet = {}
    referenceAlignmentFile = referenceFastaFile + "_Index.txt"
    if os.path.exists(referenceAlignmentFile):
        seqsAndMutatedSeqs = getFastaDictionary(referenceAlignmentFile)
        count = 0
        for name in seqsAndMutatedSeqs:
            if name in refSequences:
                count += 1
                trueSeq = seqsAndMutatedSeqs[name]
                mutatedSeq = seqsAndMutatedSeqs[name + "_mutated"]
                assert mutatedSeq == refSequences[name]
                for i in range(len(trueSeq)):
                    if trueSeq[i] != mutatedSeq[i]:
                        snpSet[(name, i)] = trueSeq[i]
            else:
                assert name.split("_")[-1] == "mutated"
        assert count == len(refSequences)
    return snpSet


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
        nullSubstitionMatrix = getNullSubstitutionMatrix()
        flatSubstitutionMatrix = getJukesCantorTypeSubstitutionMatrix()
        hmmErrorSubstitutionMatrix = loadHmmErrorSubstitutionMatrix(trainedModelPath("blasr_hmm_20.txt"))  # :56
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
                    configs = ((flatSubstitutionMatrix, expectations, seenE), (hmmErrorSubstitutionMatrix, expectations, seenE),
                               (flatSubstitutionMatrix, frequencies, seenF), (hmmErrorSubstitutionMatrix, frequencies, seenF))
                    for k, refSeqName in enumerate(refNames):
                        refSeq = refSequences[refSeqName]
                        for refPosition in range(len(refSeq)):
                            mutatedRefBase = refSeq[refPosition].upper()
                            trueRefBase = snpSet.get((refSeqName, refPosition), mutatedRefBase).upper()
                            for (errorM, table, seen), snpCalls in zip(configs, callSets):
                                if not seen[k][refPosition]:
                                    snpCalls.notCalled += 1
                                    continue
                                e = table[k][refPosition]
                                total = float(e.sum())
                                if total > 0.0 and mutatedRefBase in bases:
                                    probs = calcBasePosteriorProbs(dict(zip(bases, (e / total).tolist())), mutatedRefBase,
                                                                   nullSubstitionMatrix, errorM)
                                    for chosenBase in bases:
                                        if chosenBase != mutatedRefBase:
                                            if trueRefBase != mutatedRefBase and trueRefBase == chosenBase:
                                                snpCalls.truePositives.append((probs[chosenBase], refPosition))
                                            else:
                                                snpCalls.falsePositives.append((probs[chosenBase], refPosition))
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
            batch.close()
        sam.close()
        with open(os.path.join(self.outputDir, "marginaliseConsensus.xml"), "w") as fh:
            fh.write(prettyXml(node))
        self.finish()
        return node
