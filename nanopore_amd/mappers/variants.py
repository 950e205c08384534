"""The realigning mapper classes of the reference, by name.

Class NAMES are part of the output layout (output/analysis_<readType>/experiment_<fastq>_<fasta>_<MapperClass>/,
nanopore/pipeline.py:101-108), so every `*Chain`, `*Realign`, `*RealignEm`, `*RealignTrainedModel[20|40]`
class of nanopore/mappers/{last,last_params,lastz,lastzParams,bwa,bwa_params,blasr,blasr_params,
combinedMapper}.py exists here with the same name and the same realignSamFile arguments.
"""
from .abstractMapper import AbstractMapper


class Last(AbstractMapper):            # last.py:5
    pass


class LastParams(Last):                # last_params.py:6  (lastal -s 2 -T 0 -Q 0 -a 1)
    pass


class Lastz(AbstractMapper):           # lastz.py:5
    pass


class LastzParams(Lastz):              # lastzParams.py:9
    pass


class Bwa(AbstractMapper):             # bwa.py:4
    pass


class BwaParams(Bwa):                  # bwa_params.py:5
    pass


class Blasr(AbstractMapper):           # blasr.py:5
    pass


class BlasrParams(Blasr):              # blasr_params.py:5
    pass


class CombinedMapper(AbstractMapper):  # combinedMapper.py:10
    pass


def _variant(base, suffix, **kwargs):
    def run(self):
        base.run(self)
        if suffix == "Chain":
            self.chainSamFile()
        else:
            self.realignSamFile(**kwargs)
    return type(base.__name__ + suffix, (base,), {"run": run, "__doc__": "%s then %s(%s)" % (
        base.__name__, "chainSamFile" if suffix == "Chain" else "realignSamFile",
        ", ".join("%s=%r" % kv for kv in sorted(kwargs.items())))})


_g = globals()
for _base in (Last, LastParams, Lastz, LastzParams, Bwa, BwaParams, Blasr, BlasrParams, CombinedMapper):
    for _suffix, _kw in (("Chain", {}), ("Realign", {}), ("RealignEm", {"doEm": True}),
                         ("RealignTrainedModel", {"useTrainedModel": True})):
        _cls = _variant(_base, _suffix, **_kw)
        _g[_cls.__name__] = _cls
# quirk kept from the reference: BwaRealignEm omits doEm=True (bwa.py:22-25)
BwaRealignEm = _variant(Bwa, "RealignEm")
# the 20 % / 40 % substitution-rate models (last_params.py:30-38, blasr_params.py:27-39)
for _base in (LastParams, BlasrParams):
    for _rate in ("20", "40"):
        _cls = _variant(_base, "RealignTrainedModel" + _rate, useTrainedModel=True,
                        trainedModelFile="blasr_hmm_%s.txt" % _rate)
        _g[_cls.__name__] = _cls
del _g, _base, _suffix, _kw, _cls, _rate


# follow-on helper classes of combinedMapper.py:25-52 (run only the chain / realign step on an existing SAM)
class CombinedMapperChain2(AbstractMapper):
    def run(self):
        self.chainSamFile()


class CombinedMapperRealign2(AbstractMapper):
    def run(self):
        self.realignSamFile()


class CombinedMapperRealignEm2(AbstractMapper):
    def run(self):
        self.realignSamFile(doEm=True)


class CombinedMapperRealignTrainedModel2(AbstractMapper):
    def run(self):
        self.realignSamFile(useTrainedModel=True)
