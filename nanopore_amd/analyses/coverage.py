"""LocalCoverage / GlobalCoverage: per-alignment and aggregate coverage, identity and indel rates
(nanopore/analyses/coverage.py:10-166), same XML attribute names (consumed by metaAnalyses/coverageSummary.py).
SURVEY.md 8f next #3: post-realign statistics.  The R plots are presentation and out of scope."""
import os
import xml.etree.ElementTree as ET
from functools import reduce
from itertools import chain

import numpy

from .. import sam as pysam
from .abstractAnalysis import AbstractAnalysis
from .alignmentUncertainty import prettyXml
from .utils import AlignedPair, getFastaDictionary, getFastqDictionary, samIterator


class ReadAlignmentCoverageCounter(object):
    """Counts coverage from a pairwise alignment.  Global alignment means the entire reference and read
    sequences, trailing indels included (coverage.py:10-65)."""

    def __init__(self, readSeqName, readSeq, refSeqName, refSeq, alignedRead, globalAlignment=False):
        self.matches = 0
        self.mismatches = 0
        self.ns = 0
        self.totalReadInsertionLength = 0
        self.totalReadInsertions = 0
        self.totalReadDeletionLength = 0
        self.totalReadDeletions = 0
        self.readSeqName = readSeqName
        self.readSeq = readSeq
        self.refSeqName = refSeqName
        self.refSeq = refSeq
        self.globalAlignment = globalAlignment
        totalReadInsertionLength, totalReadDeletionLength = 0, 0
        aP = None
        for aP in AlignedPair.iterator(alignedRead, self.refSeq, self.readSeq):
            if aP.isMatch():
                self.matches += 1
            elif aP.isMismatch():
                self.mismatches += 1
            else:
                self.ns += 1
            ins = aP.getPrecedingReadInsertionLength(self.globalAlignment)
            if ins > 0:
                self.totalReadInsertions += 1
                totalReadInsertionLength += ins
            dele = aP.getPrecedingReadDeletionLength(self.globalAlignment)
            if dele > 0:
                self.totalReadDeletions += 1
                totalReadDeletionLength += dele
        if self.globalAlignment and aP is not None:  # trailing indels (coverage.py:46-61)
            assert len(self.refSeq) - aP.refPos - 1 >= 0
            if len(self.refSeq) - aP.refPos - 1 > 0:
                self.totalReadDeletions += 1
                self.totalReadDeletionLength += len(self.refSeq) - aP.refPos - 1
            if alignedRead.is_reverse:
                if aP.readPos > 0:
                    self.totalReadInsertions += 1
                    totalReadInsertionLength += aP.readPos
            else:
                assert len(self.readSeq) - aP.readPos - 1 >= 0
                if len(self.readSeq) - aP.readPos - 1 > 0:
                    self.totalReadInsertions += 1
                    totalReadInsertionLength += len(self.readSeq) - aP.readPos - 1
        assert totalReadInsertionLength <= len(self.readSeq)
        assert totalReadDeletionLength <= len(self.refSeq)
        self.totalReadInsertionLength += totalReadInsertionLength
        self.totalReadDeletionLength += totalReadDeletionLength

    def readCoverage(self):
        return AbstractAnalysis.formatRatio(self.matches + self.mismatches,
                                            self.matches + self.mismatches + self.totalReadInsertionLength)

    def referenceCoverage(self):
        return AbstractAnalysis.formatRatio(self.matches + self.mismatches,
                                            self.matches + self.mismatches + self.totalReadDeletionLength)

    def identity(self):
        return AbstractAnalysis.formatRatio(self.matches, self.matches + self.mismatches + self.totalReadInsertionLength)

    def mismatchesPerReadBase(self):
        return AbstractAnalysis.formatRatio(self.mismatches, self.matches + self.mismatches)

    def deletionsPerReadBase(self):
        return AbstractAnalysis.formatRatio(self.totalReadDeletions, self.matches + self.mismatches)

    def insertionsPerReadBase(self):
        return AbstractAnalysis.formatRatio(self.totalReadInsertions, self.matches + self.mismatches)

    def readLength(self):
        return len(self.readSeq)

    def getXML(self):
        return ET.Element("readAlignmentCoverage", {
            "refSeqName": self.refSeqName, "readSeqName": self.readSeqName, "readLength": str(self.readLength()),
            "readCoverage": str(self.readCoverage()), "referenceCoverage": str(self.referenceCoverage()),
            "identity": str(self.identity()), "mismatchesPerReadBase": str(self.mismatchesPerReadBase()),
            "insertionsPerReadBase": str(self.insertionsPerReadBase()),
            "deletionsPerReadBase": str(self.deletionsPerReadBase())})


def getAggregateCoverageStats(readAlignmentCoverages, tagName, refSequences, readSequences, readsToReadAlignmentCoverages,
                              typeof):
    """Aggregate stats across a set of read alignments (coverage.py:97-125)."""
    if typeof == "coverage_all":
        mappedReadLengths = list(chain(*[[len(readSequences[i])] * len(readsToReadAlignmentCoverages[i])
                                         for i in readSequences if i in readsToReadAlignmentCoverages]))
    else:
        mappedReadLengths = [len(readSequences[i]) for i in readSequences if i in readsToReadAlignmentCoverages]
    unmappedReadLengths = [len(readSequences[i]) for i in readSequences if i not in readsToReadAlignmentCoverages]

    def stats(fnStringName):
        values = [getattr(x, fnStringName)() for x in readAlignmentCoverages]
        ordered = sorted(values)
        return ordered[0], numpy.average(ordered), numpy.median(ordered), ordered[-1], " ".join(map(str, values))

    attribs = {"numberOfReadAlignments": str(len(readAlignmentCoverages)), "numberOfReads": str(len(readSequences)),
               "numberOfReferenceSequences": str(len(refSequences)), "numberOfMappedReads": str(len(mappedReadLengths)),
               "mappedReadLengths": " ".join(map(str, mappedReadLengths)),
               "numberOfUnmappedReads": str(len(unmappedReadLengths)),
               "unmappedReadLengths": " ".join(map(str, unmappedReadLengths))}
    for fnStringName in ("readCoverage", "referenceCoverage", "identity", "mismatchesPerReadBase", "deletionsPerReadBase",
                         "insertionsPerReadBase", "readLength"):
        for prefix, value in zip(("min", "avg", "median", "max", "distribution"), stats(fnStringName)):
            attribs[prefix + fnStringName] = str(value)
    parentNode = ET.Element(tagName, attribs)
    for c in readAlignmentCoverages:
        parentNode.append(c.getXML())
    return parentNode


class LocalCoverage(AbstractAnalysis):
    """Calculates coverage, treating alignments as local alignments (coverage.py:127-160)."""

    def run(self, globalAlignment=False):
        AbstractAnalysis.run(self)
        refSequences = getFastaDictionary(self.referenceFastaFile)
        readSequences = getFastqDictionary(self.readFastqFile)
        sam = pysam.Samfile(self.samFile, "r")
        readsToReadCoverages = {}
        for aR in samIterator(sam):
            refName = sam.getrname(aR.rname)
            counter = ReadAlignmentCoverageCounter(aR.qname, readSequences[aR.qname], refName, refSequences[refName], aR,
                                                   globalAlignment)
            readsToReadCoverages.setdefault(aR.qname, []).append(counter)
        sam.close()
        if readsToReadCoverages:
            everything = reduce(lambda x, y: x + y, readsToReadCoverages.values())
            best = [max(x, key=lambda y: y.readCoverage()) for x in readsToReadCoverages.values()]
            for readCoverages, outputName in ((everything, "coverage_all"), (best, "coverage_bestPerRead")):
                parentNode = getAggregateCoverageStats(readCoverages, outputName, refSequences, readSequences,
                                                       readsToReadCoverages, outputName)
                with open(os.path.join(self.outputDir, outputName + ".xml"), "w") as fh:
                    fh.write(prettyXml(parentNode))
                with open(os.path.join(self.outputDir, outputName + ".txt"), "w") as outf:
                    outf.write("MappedReadLengths " + parentNode.get("mappedReadLengths") + "\n")
                    outf.write("UnmappedReadLengths " + parentNode.get("unmappedReadLengths") + "\n")
                    outf.write("ReadCoverage " + parentNode.get("distributionreadCoverage") + "\n")
                    outf.write("MismatchesPerReadBase " + parentNode.get("distributionmismatchesPerReadBase") + "\n")
                    outf.write("ReadIdentity " + parentNode.get("distributionidentity") + "\n")
                    outf.write("InsertionsPerBase " + parentNode.get("distributioninsertionsPerReadBase") + "\n")
                    outf.write("DeletionsPerBase " + parentNode.get("distributiondeletionsPerReadBase") + "\n")
        self.finish()


class GlobalCoverage(LocalCoverage):
    """Coverage treating alignments as global alignments (coverage.py:162-166)."""

    def run(self):
        LocalCoverage.run(self, globalAlignment=True)
