"""AbstractMapper with the reference's constructor and helper signatures
(nanopore/mappers/abstractMapper.py:10-39).  The external base mappers (last, bwa, lastz, blasr) are out
of scope and absent from the snapshot: `run()` of a base mapper here only checks that
their SAM output exists.  `chainSamFile()` and `realignSamFile()` -- the entry into the hot path -- are real.
"""
import os
import shutil

from ..analyses.utils import chainSamFile, realignSamFileTargetFn, trainedModelPath
from ..bioio import Target


class AbstractMapper(Target):
    """Base class for mappers.  Inherit this class to create a mapper."""

    def __init__(self, readFastqFile, readType, referenceFastaFile, outputSamFile, emptyHmmFile=None):
        Target.__init__(self)
        self.readFastqFile = readFastqFile
        self.referenceFastaFile = referenceFastaFile
        self.outputSamFile = outputSamFile
        self.readType = readType
        self.emptyHmmFile = emptyHmmFile

    def run(self, params=""):
        """The external mapper has to have written self.outputSamFile (its command line is outside this build)."""
        if not os.path.exists(self.outputSamFile):
            raise RuntimeError("%s: %s does not exist; run the external mapper first (base mappers are out of scope of "
                               "this build)" % (self.__class__.__name__, self.outputSamFile))

    def chainSamFile(self):
        """Converts the sam file so that there is at most one global alignment of each read
        (abstractMapper.py:18-23)."""
        tempSamFile = os.path.join(self.getLocalTempDir(), "temp.sam")
        shutil.copyfile(self.outputSamFile, tempSamFile)
        chainSamFile(tempSamFile, self.outputSamFile, self.readFastqFile, self.referenceFastaFile)

    def selectHmmFile(self, doEm=False, useTrainedModel=False, trainedModelFile="blasr_hmm_0.txt"):
        """Model selection truth table of abstractMapper.py:29-37."""
        if useTrainedModel and doEm:
            raise RuntimeError("Attempting to train stock model")
        if doEm:
            return self.emptyHmmFile
        if useTrainedModel:
            return trainedModelPath(trainedModelFile)
        return None

    def realignSamFile(self, gapGamma=0.5, matchGamma=0.0, doEm=False, useTrainedModel=False,
                       trainedModelFile="blasr_hmm_0.txt"):
        """Chains and then realigns the resulting global alignments (abstractMapper.py:25-39)."""
        hmmFile = self.selectHmmFile(doEm, useTrainedModel, trainedModelFile)
        tempSamFile = os.path.join(self.getGlobalTempDir(), "temp_in.sam")
        shutil.copyfile(self.outputSamFile, tempSamFile)
        return realignSamFileTargetFn(self, tempSamFile, self.outputSamFile, self.readFastqFile,
                                      self.referenceFastaFile, gapGamma, matchGamma, hmmFile, doEm)
