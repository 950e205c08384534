"""Substitutions: nucleotide substitution matrix over all aligned pairs (nanopore/analyses/substitutions.py:9-82)."""
import os
import xml.etree.ElementTree as ET

from .. import sam as pysam
from .abstractAnalysis import AbstractAnalysis
from .alignmentUncertainty import prettyXml
from .utils import AlignedPair, getFastaDictionary, getFastqDictionary, samIterator


class SubstitutionMatrix(object):
    """Nucleotide substitution counts, with a fifth row / column for wildcards (substitutions.py:9-56)."""

    def __init__(self):
        self.matrix = [0.0] * 25

    @staticmethod
    def _index(base):
        base = base.upper()
        return {"A": 0, "C": 1, "G": 2, "T": 3}.get(base, 4)

    def addAlignedPair(self, refBase, readBase):
        self.matrix[self._index(refBase) * 5 + self._index(readBase)] += 1

    def getCount(self, refBase, readBase):
        return self.matrix[self._index(refBase) * 5 + self._index(readBase)]

    def getFreqs(self, refBase, bases):
        freqs = [self.getCount(refBase, b) for b in bases]
        if sum(freqs) == 0:
            return [0.0] * len(freqs)
        return [x / sum(freqs) for x in freqs]

    def getXML(self):
        def _identity(matches, mismatches):
            if matches + mismatches == 0:
                return "NaN"
            return matches / (mismatches + matches)
        matches = sum(self.getCount(b, b) for b in "ACTG")
        mismatches = sum(sum(self.getCount(r, q) for q in "ACTG" if q != r) for r in "ACTG")
        node = ET.Element("substitutions", {"matches": str(matches), "mismatches": str(mismatches),
                                            "identity": str(_identity(matches, mismatches))})
        for refBase in "ACGTN":
            matches = self.getCount(refBase, refBase)
            mismatches = sum(self.getCount(refBase, q) for q in "ACTG" if q != refBase)
            baseNode = ET.SubElement(node, refBase, {"matches": str(matches), "mismatches": str(mismatches),
                                                     "identity": str(_identity(matches, mismatches))})
            for readBase in "ACGTN":
                ET.SubElement(baseNode, readBase, {"count": str(self.getCount(refBase, readBase))})
        return node


class Substitutions(AbstractAnalysis):
    def run(self, kmer=5):
        AbstractAnalysis.run(self)
        refSequences = getFastaDictionary(self.referenceFastaFile)
        readSequences = getFastqDictionary(self.readFastqFile)
        sM = SubstitutionMatrix()
        sam = pysam.Samfile(self.samFile, "r")
        for aR in samIterator(sam):
            for aP in AlignedPair.iterator(aR, refSequences[sam.getrname(aR.rname)], readSequences[aR.qname]):
                sM.addAlignedPair(aP.getRefBase(), aP.getReadBase())
        sam.close()
        with open(os.path.join(self.outputDir, "substitutions.xml"), "w") as fh:
            fh.write(prettyXml(sM.getXML()))
        with open(os.path.join(self.outputDir, "subst.tsv"), "w") as outf:
            outf.write("A\tC\tG\tT\n")
            for x in "ACGT":
                outf.write("{}\t{}\n".format(x, "\t".join(map(str, sM.getFreqs(x, "ACGT")))))
        self.finish()
        return sM
