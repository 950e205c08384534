"""Substitutions: substitutions.xml and subst.tsv from the 5 x 5 counts of aligned (reference base, read base).

Schema of nanopore/analyses/substitutions.py:34-50 -- root <substitutions matches mismatches identity>, one child per
reference base A C G T N with the same three attributes, under it one child per read base with its count -- and the
row-normalised 4 x 4 table of :66-72.  Counts are floats in the reference (its matrix starts as 0.0), hence "12.0".  The
counting itself is the device table of alignmentStats.SamAlignmentStats summed over the records."""
import os
import xml.etree.ElementTree as ET

from .abstractAnalysis import AbstractAnalysis
from .alignmentStats import SamAlignmentStats
from .alignmentUncertainty import prettyXml

_BASES = "ACGTN"


class SubstitutionMatrix(object):
    """25 counts, reference base major; anything outside ACGT is N."""

    def __init__(self, counts=None):
        self.matrix = [0.0] * 25 if counts is None else [float(v) for v in counts]

    @staticmethod
    def _index(base):
        return "ACGT".find(base.upper()) % 5 if base.upper() in "ACGT" else 4

    def addAlignedPair(self, refBase, readBase):
        self.matrix[5 * self._index(refBase) + self._index(readBase)] += 1

    def getCount(self, refBase, readBase):
        return self.matrix[5 * self._index(refBase) + self._index(readBase)]

    def getFreqs(self, refBase, bases):
        row = [self.getCount(refBase, b) for b in bases]
        total = sum(row)
        return [v / total for v in row] if total else [0.0] * len(row)

    def _tally(self, refBases):
        same = sum(self.getCount(b, b) for b in refBases)
        other = sum(self.getCount(b, q) for b in refBases for q in "ACGT" if q != b)
        return same, other, (same / (same + other) if same + other else "NaN")

    def getXML(self):
        root = ET.Element("substitutions", dict(zip(("matches", "mismatches", "identity"), map(str, self._tally("ACGT")))))
        for refBase in _BASES:
            node = ET.SubElement(root, refBase, dict(zip(("matches", "mismatches", "identity"), map(str, self._tally(refBase)))))
            for readBase in _BASES:
                ET.SubElement(node, readBase, {"count": str(self.getCount(refBase, readBase))})
        return root


class Substitutions(AbstractAnalysis):
    def run(self, kmer=5, ctx=None, stats=None):
        AbstractAnalysis.run(self)
        stats = stats or SamAlignmentStats(self.samFile, self.referenceFastaFile, self.readFastqFile, ctx=ctx)
        sM = SubstitutionMatrix(stats.substitutionCounts().reshape(-1))
        with open(os.path.join(self.outputDir, "substitutions.xml"), "w") as fh:
            fh.write(prettyXml(sM.getXML()))
        with open(os.path.join(self.outputDir, "subst.tsv"), "w") as fh:
            fh.write("A\tC\tG\tT\n")
            for refBase in "ACGT":
                fh.write("%s\t%s\n" % (refBase, "\t".join(str(f) for f in sM.getFreqs(refBase, "ACGT"))))
        self.finish()
        return sM
