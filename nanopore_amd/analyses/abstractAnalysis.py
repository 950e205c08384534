"""Base class of the analyses: the constructor signature the pipeline instantiates them with
(nanopore/pipeline.py:133: readFastqFile, readType, referenceFastaFile, samFile, outputDir), the `DONE` marker file
that keeps a finished analysis from being repeated, and the nan-on-zero ratio every XML attribute goes through
(interface of nanopore/analyses/abstractAnalysis.py:8-41)."""
import os

from ..bioio import Target

_MARKER = "DONE"


class AbstractAnalysis(Target):
    def __init__(self, readFastqFile, readType, referenceFastaFile, samFile, outputDir):
        Target.__init__(self)
        self.readFastqFile, self.readType = readFastqFile, readType
        self.referenceFastaFile, self.samFile, self.outputDir = referenceFastaFile, samFile, outputDir

    def run(self):
        """Subclasses do the work; the reference only logs its inputs here."""

    def finish(self):
        with open(os.path.join(self.outputDir, _MARKER), "w"):
            pass

    @staticmethod
    def isFinished(outputDir):
        return os.path.isfile(os.path.join(outputDir, _MARKER))

    @staticmethod
    def reset(outputDir):
        try:
            os.remove(os.path.join(outputDir, _MARKER))
        except FileNotFoundError:
            pass

    @staticmethod
    def formatRatio(numerator, denominator):
        try:
            return float(numerator) / denominator
        except ZeroDivisionError:
            return float("nan")
