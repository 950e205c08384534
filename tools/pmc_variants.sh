#!/bin/bash
# tools/pmc_variants.sh OUT tag... -- two rocprofv3 --pmc passes (instruction counts; busy / wait cycles) of the k_dp_rs<2> launches of
# tools/gpu_pmc_run.py 12288 10000 200 for library variants (NPR_LIB), one summary line per variant.  Bring-up tool.
set -u
OUT=gpurun_out/$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  lib=$R/nanopore_amd/libnprealign.so; [ "$tag" != default ] && lib=$R/nanopore_amd/libnprealign_$tag.so
  export NPR_LIB=$lib
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --output-format csv -d $R/$OUT/$tag/g0 -- python $R/tools/gpu_pmc_run.py 12288 10000 200 > $R/$OUT/$tag.g0.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/$tag/g1 -- python $R/tools/gpu_pmc_run.py 12288 10000 200 > $R/$OUT/$tag.g1.log 2>&1
done
python3 - "$R/$OUT" "$@" <<'PY'
import csv, glob, collections, sys
out, tags = sys.argv[1], sys.argv[2:]
for tag in tags:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("%s/%s/g*/**/*counter_collection.csv" % (out, tag), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_dp_rs" in row["Kernel_Name"]:
                per[row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
    m = {k: sum(v.values()) / len(v) for k, v in per.items()}
    ms = [l for l in open("%s/%s.g1.log" % (out, tag)) if l.startswith("cells")]
    print(tag, " ".join("%s=%.4g" % (k.replace("SQ_", ""), m[k]) for k in sorted(m)), ms[-1].strip() if ms else "")
PY
