"""AbstractAnalysis with the reference's constructor, DONE marker and ratio helper
(nanopore/analyses/abstractAnalysis.py:8-41)."""
import os

from ..bioio import Target


class AbstractAnalysis(Target):
    def __init__(self, readFastqFile, readType, referenceFastaFile, samFile, outputDir):
        Target.__init__(self)
        self.readFastqFile = readFastqFile
        self.referenceFastaFile = referenceFastaFile
        self.samFile = samFile
        self.outputDir = outputDir
        self.readType = readType

    def run(self):
        pass

    def finish(self):
        """Marks the analysis as done so that it is not repeated (abstractAnalysis.py:23-26)."""
        open(os.path.join(self.outputDir, "DONE"), "w").close()

    @staticmethod
    def reset(outputDir):
        if AbstractAnalysis.isFinished(outputDir):
            os.remove(os.path.join(outputDir, "DONE"))

    @staticmethod
    def isFinished(outputDir):
        return os.path.exists(os.path.join(outputDir, "DONE"))

    @staticmethod
    def formatRatio(numerator, denominator):
        if denominator == 0:
            return float("nan")
        return float(numerator) / denominator
