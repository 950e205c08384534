"""AlignmentUncertainty: average posterior match probability of the alignments in a SAM file
(nanopore/analyses/alignmentUncertainty.py:12-71).

The reference runs, per record and serially, `cactus_realign --rescoreByPosteriorProbIgnoringGaps
--rescoreOriginalAlignment --diagonalExpansion=10 --splitMatrixBiggerThanThis=100 --loadHmm=blasr_hmm_0.txt`
(:41) and reads back the cigar's score (:48).  Here every record goes to the GPU in one batched
NPR_MODE_RESCORE_ORIGINAL call; the XML has the same root and attributes (:59-64).
"""
import os
import xml.etree.ElementTree as ET
from xml.dom import minidom

from .. import sam as pysam
from .abstractAnalysis import AbstractAnalysis
from .utils import (ANALYSIS_SPLIT_MATRIX_BIGGER_THAN, getFastaDictionary, realignRecords, samIterator,
                    trainedModelPath)


def prettyXml(elem):
    return minidom.parseString(ET.tostring(elem, "utf-8")).toprettyxml(indent="  ")


class AlignmentUncertainty(AbstractAnalysis):
    def run(self, ctx=None):
        from .. import realign
        AbstractAnalysis.run(self)
        refSequences = getFastaDictionary(self.referenceFastaFile)
        sam = pysam.Samfile(self.samFile, "r")
        records = list(samIterator(sam))
        hmmFile = trainedModelPath("blasr_hmm_0.txt")                         # alignmentUncertainty.py:38
        results = realignRecords(sam, records, refSequences, 0.5, 0.0, hmmFile, mode=realign.MODE_RESCORE_ORIGINAL,
                                 splitThreshold=ANALYSIS_SPLIT_MATRIX_BIGGER_THAN, ctx=ctx)
        avgPosteriorMatchProbabilityInCigar = []
        alignedPairsInCigar = []
        for aR, r in zip(records, results):
            if r["status"] != 0:
                raise RuntimeError("Rescoring failed for %s: status %d" % (aR.qname, r["status"]))
            avgPosteriorMatchProbabilityInCigar.append(r["score"])            # pA.score, :48
            matches = sum(length for op, length in r["ops"] if op == 0)
            alignedPairsInCigar.append(matches)
            # rescoring keeps the original alignment's pairs (:51-52)
            assert matches == sum(1 for q, t in aR.aligned_pairs if q is not None and t is not None)
        sam.close()
        node = ET.Element("alignmentUncertainty", {
            "averagePosteriorMatchProbabilityPerRead": str(self.formatRatio(sum(avgPosteriorMatchProbabilityInCigar),
                                                                            len(avgPosteriorMatchProbabilityInCigar))),
            "averagePosteriorMatchProbability": str(self.formatRatio(
                float(sum(p * n for p, n in zip(avgPosteriorMatchProbabilityInCigar, alignedPairsInCigar))),
                sum(alignedPairsInCigar))),
            "averagePosteriorMatchProbabilitesPerRead": ",".join(str(i) for i in avgPosteriorMatchProbabilityInCigar),
            "alignedPairsInCigar": ",".join(str(i) for i in alignedPairsInCigar)})
        with open(os.path.join(self.outputDir, "alignmentUncertainty.xml"), "w") as fh:
            fh.write(prettyXml(node))
        self.finish()
        return node
