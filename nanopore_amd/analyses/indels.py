"""Indels: read insertion / deletion length statistics per alignment and in aggregate
(nanopore/analyses/indels.py:9-110)."""
import os
import xml.etree.ElementTree as ET
from functools import reduce

import numpy

from .. import sam as pysam
from .abstractAnalysis import AbstractAnalysis
from .alignmentUncertainty import prettyXml
from .utils import AlignedPair, getFastaDictionary, getFastqDictionary, samIterator


def _avg(values):
    return numpy.average(values) if len(values) else float("nan")


def _median(values):
    return numpy.median(values) if len(values) else float("nan")


class IndelCounter(object):
    def __init__(self, refSeqName, refSeq, readSeqName, readSeq, alignedRead):
        self.readInsertionLengths = []
        self.readDeletionLengths = []
        self.blockLengths = []
        self.readSeqName = readSeqName
        self.readSeq = readSeq
        self.refSeqName = refSeqName
        self.refSeq = refSeq
        blockLength = 0
        for aP in AlignedPair.iterator(alignedRead, self.refSeq, self.readSeq):
            ins, dele = aP.getPrecedingReadInsertionLength(), aP.getPrecedingReadDeletionLength()
            if ins > 0:
                self.readInsertionLengths.append(ins)
            if dele > 0:
                self.readDeletionLengths.append(dele)
            if ins > 0 or dele > 0:
                assert blockLength > 0
                self.blockLengths.append(blockLength)
                blockLength = 1
            else:
                blockLength += 1

    def getXML(self):
        return ET.Element("indels", {
            "refSeqName": self.refSeqName, "refSeqLength": str(len(self.refSeq)), "readSeqName": self.readSeqName,
            "readSeqLength": str(len(self.readSeq)), "numberReadInsertions": str(len(self.readInsertionLengths)),
            "numberReadDeletions": str(len(self.readDeletionLengths)),
            "avgReadInsertionLength": str(_avg(self.readInsertionLengths)),
            "avgReadDeletionLength": str(_avg(self.readDeletionLengths)),
            "medianReadInsertionLength": str(_median(self.readInsertionLengths)),
            "medianReadDeletionLength": str(_median(self.readDeletionLengths)),
            "readInsertionLengths": " ".join(str(i) for i in self.readInsertionLengths),
            "readDeletionLengths": " ".join(str(i) for i in self.readDeletionLengths)})


def getAggregateIndelStats(indelCounters):
    """Aggregate stats across a set of read alignments (indels.py:47-82).  As in the reference, each of the five
    per-alignment distributions ends up under its bare name holding the sorted distribution string (the min / avg /
    median / max values are computed and then overwritten there, indels.py:77-78); indels.tsv is built from these."""
    readInsertionLengths = reduce(lambda x, y: x + y, [ic.readInsertionLengths for ic in indelCounters])
    readDeletionLengths = reduce(lambda x, y: x + y, [ic.readDeletionLengths for ic in indelCounters])
    attribs = {"numberOfReadAlignments": str(len(indelCounters)),
               "readInsertionLengths": " ".join(map(str, readInsertionLengths)),
               "readDeletionLengths": " ".join(map(str, readDeletionLengths))}
    for name, distribution in (("ReadSequenceLengths", [len(ic.readSeq) for ic in indelCounters]),
                               ("NumberReadInsertions", [len(ic.readInsertionLengths) for ic in indelCounters]),
                               ("NumberReadDeletions", [len(ic.readDeletionLengths) for ic in indelCounters]),
                               ("MedianReadInsertionLengths", [_median(ic.readInsertionLengths) for ic in indelCounters]),
                               ("MedianReadDeletionLengths", [_median(ic.readDeletionLengths) for ic in indelCounters])):
        attribs[name] = " ".join(map(str, sorted(distribution)))
    parentNode = ET.Element("indels", attribs)
    for ic in indelCounters:
        parentNode.append(ic.getXML())
    return parentNode


class Indels(AbstractAnalysis):
    def run(self):
        AbstractAnalysis.run(self)
        refSequences = getFastaDictionary(self.referenceFastaFile)
        readSequences = getFastqDictionary(self.readFastqFile)
        sam = pysam.Samfile(self.samFile, "r")
        indelCounters = [IndelCounter(sam.getrname(aR.rname), refSequences[sam.getrname(aR.rname)], aR.qname,
                                      readSequences[aR.qname], aR) for aR in samIterator(sam)]
        sam.close()
        if indelCounters:
            indelXML = getAggregateIndelStats(indelCounters)
            with open(os.path.join(self.outputDir, "indels.xml"), "w") as fh:
                fh.write(prettyXml(indelXML))
            var = ["readInsertionLengths", "readDeletionLengths", "ReadSequenceLengths", "NumberReadInsertions",
                   "NumberReadDeletions", "MedianReadInsertionLengths", "MedianReadDeletionLengths"]
            columns = [[x] + indelXML.attrib[x].split() for x in var]
            depth = max(len(c) for c in columns)
            with open(os.path.join(self.outputDir, "indels.tsv"), "w") as tmp:
                for i in range(depth):  # transposed, short columns padded with None like Python 2's map(None, ...)
                    tmp.write("\t".join(str(c[i]) if i < len(c) else "None" for c in columns) + "\n")
        self.finish()
