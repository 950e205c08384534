"""LocalCoverage / GlobalCoverage: coverage_all.xml, coverage_bestPerRead.xml and their .txt tables.

Schema of nanopore/analyses/coverage.py: one <readAlignmentCoverage> per SAM record (attributes :79-86: refSeqName,
readSeqName, readLength, readCoverage, referenceCoverage, identity, mismatchesPerReadBase, insertionsPerReadBase,
deletionsPerReadBase, with the ratios defined at :60-77) under a root that carries the read / reference counts, the
mapped / unmapped read lengths and min / avg / median / max / distribution of each of the seven quantities (:88-125;
read by metaAnalyses/coverageSummary.py:67-73).  The counts behind the ratios come from the device table of
alignmentStats.SamAlignmentStats; the R plot of the reference (:153) is outside this build's scope."""
import os
import xml.etree.ElementTree as ET

import numpy as np

from .abstractAnalysis import AbstractAnalysis
from .alignmentStats import MATCHES, MISMATCHES, SamAlignmentStats
from .alignmentUncertainty import prettyXml

QUANTITIES = ("readCoverage", "referenceCoverage", "identity", "mismatchesPerReadBase", "deletionsPerReadBase",
              "insertionsPerReadBase", "readLength")


def _ratio(a, b):
    return AbstractAnalysis.formatRatio(int(a), int(b))


def coverageColumns(stats, globalAlignment):
    """The seven per-record quantities (coverage.py:60-77) as a dict of lists, from the integer table."""
    m, x = stats.table[:, MATCHES].astype(np.int64), stats.table[:, MISMATCHES].astype(np.int64)
    n_ins, ins_len, n_del, del_len = stats.indelTotals(globalAlignment)
    aligned = m + x
    n = len(stats)
    return {
        "readCoverage": [_ratio(aligned[i], aligned[i] + ins_len[i]) for i in range(n)],
        "referenceCoverage": [_ratio(aligned[i], aligned[i] + del_len[i]) for i in range(n)],
        "identity": [_ratio(m[i], aligned[i] + ins_len[i]) for i in range(n)],
        "mismatchesPerReadBase": [_ratio(x[i], aligned[i]) for i in range(n)],
        "deletionsPerReadBase": [_ratio(n_del[i], aligned[i]) for i in range(n)],
        "insertionsPerReadBase": [_ratio(n_ins[i], aligned[i]) for i in range(n)],
        "readLength": [int(v) for v in stats.readLength],
    }


def aggregateNode(tagName, stats, columns, chosen, perAlignmentLengths):
    """Root element for the records `chosen` (indices into the table)."""
    aligned_reads = {}
    for i in range(len(stats)):
        aligned_reads.setdefault(stats.readNames[i], 0)
        aligned_reads[stats.readNames[i]] += 1
    mapped, unmapped = [], []
    for name, seq in stats.readSequences.items():
        if name in aligned_reads:
            mapped.extend([len(seq)] * (aligned_reads[name] if perAlignmentLengths else 1))
        else:
            unmapped.append(len(seq))
    attrib = {"numberOfReadAlignments": str(len(chosen)), "numberOfReads": str(len(stats.readSequences)),
              "numberOfReferenceSequences": str(len(stats.refSequences)), "numberOfMappedReads": str(len(mapped)),
              "mappedReadLengths": " ".join(map(str, mapped)), "numberOfUnmappedReads": str(len(unmapped)),
              "unmappedReadLengths": " ".join(map(str, unmapped))}
    for q in QUANTITIES:
        values = [columns[q][i] for i in chosen]
        ordered = sorted(values)
        attrib["min" + q], attrib["max" + q] = str(ordered[0]), str(ordered[-1])
        attrib["avg" + q], attrib["median" + q] = str(np.average(ordered)), str(np.median(ordered))
        attrib["distribution" + q] = " ".join(map(str, values))
    root = ET.Element(tagName, attrib)
    for i in chosen:
        ET.SubElement(root, "readAlignmentCoverage", {
            "refSeqName": stats.refNames[i], "readSeqName": stats.readNames[i], "readLength": str(columns["readLength"][i]),
            "readCoverage": str(columns["readCoverage"][i]), "referenceCoverage": str(columns["referenceCoverage"][i]),
            "identity": str(columns["identity"][i]), "mismatchesPerReadBase": str(columns["mismatchesPerReadBase"][i]),
            "insertionsPerReadBase": str(columns["insertionsPerReadBase"][i]),
            "deletionsPerReadBase": str(columns["deletionsPerReadBase"][i])})
    return root


class LocalCoverage(AbstractAnalysis):
    """Coverage with every record taken as a local alignment."""

    def run(self, globalAlignment=False, ctx=None, stats=None):
        """`stats`: a SamAlignmentStats made elsewhere (SamAlignmentStats.fromRealignedSam: the table a realignment job reduced
        on the device); default: the records of self.samFile are counted now."""
        AbstractAnalysis.run(self)
        stats = stats or SamAlignmentStats(self.samFile, self.referenceFastaFile, self.readFastqFile, ctx=ctx)
        if len(stats):
            columns = coverageColumns(stats, globalAlignment)
            # the record with the highest read coverage of every read, reads in order of first appearance (:139)
            best, order = {}, []
            for i, name in enumerate(stats.readNames):
                if name not in best:
                    best[name] = i
                    order.append(name)
                elif columns["readCoverage"][i] > columns["readCoverage"][best[name]]:
                    best[name] = i
            # coverage_all lists the records read by read, as the reference's dict of per-read lists does
            by_read = [i for name in order for i in range(len(stats)) if stats.readNames[i] == name] if len(order) < len(stats) \
                else list(range(len(stats)))
            for tag, chosen, perAlignment in (("coverage_all", by_read, True), ("coverage_bestPerRead", [best[k] for k in order], False)):
                node = aggregateNode(tag, stats, columns, chosen, perAlignment)
                with open(os.path.join(self.outputDir, tag + ".xml"), "w") as fh:
                    fh.write(prettyXml(node))
                with open(os.path.join(self.outputDir, tag + ".txt"), "w") as fh:   # one quantity per line, variable length (:143-151)
                    for label, key in (("MappedReadLengths", "mappedReadLengths"), ("UnmappedReadLengths", "unmappedReadLengths"),
                                       ("ReadCoverage", "distributionreadCoverage"),
                                       ("MismatchesPerReadBase", "distributionmismatchesPerReadBase"),
                                       ("ReadIdentity", "distributionidentity"), ("InsertionsPerBase", "distributioninsertionsPerReadBase"),
                                       ("DeletionsPerBase", "distributiondeletionsPerReadBase")):
                        fh.write("%s %s\n" % (label, node.get(key)))
        self.finish()


class GlobalCoverage(LocalCoverage):
    """Coverage with every record taken as a global alignment: unaligned ends count as indels."""

    def run(self, ctx=None, stats=None):
        LocalCoverage.run(self, globalAlignment=True, ctx=ctx, stats=stats)
