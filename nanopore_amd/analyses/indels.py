"""Indels: indels.xml and indels.tsv.

Schema of nanopore/analyses/indels.py: one <indels> element per SAM record (:33-45: names, lengths, number / average /
median / list of read insertion and deletion lengths) under a root <indels> with the concatenated lists and, per
aggregate distribution, its sorted values (:47-82 -- the reference's loop leaves only the distribution string under the
bare name, which is what its TSV writer then reads at :96-107); the TSV is those seven lists side by side, ragged
columns padded with "None".  Gap lengths come from alignmentStats.SamAlignmentStats (the number of insertions / deletions
per record is cross-checked against the device table)."""
import os
import xml.etree.ElementTree as ET
from itertools import zip_longest

import numpy as np

from .abstractAnalysis import AbstractAnalysis
from .alignmentStats import N_DEL, N_INS, SamAlignmentStats
from .alignmentUncertainty import prettyXml

TSV_COLUMNS = ("readInsertionLengths", "readDeletionLengths", "ReadSequenceLengths", "NumberReadInsertions",
               "NumberReadDeletions", "MedianReadInsertionLengths", "MedianReadDeletionLengths")


def _text(values):
    return " ".join(str(v) for v in values)


def recordNode(stats, i, ins, dels):
    return ET.Element("indels", {
        "refSeqName": stats.refNames[i], "refSeqLength": str(int(stats.refLength[i])), "readSeqName": stats.readNames[i],
        "readSeqLength": str(int(stats.readLength[i])), "numberReadInsertions": str(len(ins)), "numberReadDeletions": str(len(dels)),
        "avgReadInsertionLength": str(np.average(ins) if ins else float("nan")),
        "avgReadDeletionLength": str(np.average(dels) if dels else float("nan")),
        "medianReadInsertionLength": str(np.median(ins) if ins else float("nan")),
        "medianReadDeletionLength": str(np.median(dels) if dels else float("nan")),
        "readInsertionLengths": _text(ins), "readDeletionLengths": _text(dels)})


def getAggregateIndelStats(stats):
    ins, dels = stats.gapLengths()
    n = len(stats)
    assert [len(v) for v in ins] == [int(v) for v in stats.table[:, N_INS]] and [len(v) for v in dels] == [int(v) for v in stats.table[:, N_DEL]]
    attrib = {"numberOfReadAlignments": str(n), "readInsertionLengths": _text(v for per in ins for v in per),
              "readDeletionLengths": _text(v for per in dels for v in per)}
    median = lambda v: float(np.median(v)) if v else float("nan")  # noqa: E731
    for name, values in (("ReadSequenceLengths", [int(v) for v in stats.readLength]), ("NumberReadInsertions", [len(v) for v in ins]),
                         ("NumberReadDeletions", [len(v) for v in dels]), ("MedianReadInsertionLengths", [median(v) for v in ins]),
                         ("MedianReadDeletionLengths", [median(v) for v in dels])):
        attrib[name] = _text(sorted(values))
    root = ET.Element("indels", attrib)
    for i in range(n):
        root.append(recordNode(stats, i, ins[i], dels[i]))
    return root


class Indels(AbstractAnalysis):
    def run(self, ctx=None, stats=None):
        """`stats`: a SamAlignmentStats made elsewhere (SamAlignmentStats.fromRealignedSam: the table a realignment job reduced
        on the device); default: the records of self.samFile are counted now."""
        AbstractAnalysis.run(self)
        stats = stats or SamAlignmentStats(self.samFile, self.referenceFastaFile, self.readFastqFile, ctx=ctx)
        if len(stats):
            root = getAggregateIndelStats(stats)
            with open(os.path.join(self.outputDir, "indels.xml"), "w") as fh:
                fh.write(prettyXml(root))
            columns = [[name] + root.attrib[name].split() for name in TSV_COLUMNS]
            with open(os.path.join(self.outputDir, "indels.tsv"), "w") as fh:
                for row in zip_longest(*columns):
                    fh.write("\t".join(str(v) for v in row) + "\n")
        self.finish()
