#!/bin/bash
# tools/pmc_cmd.sh OUT KERNEL_SUBSTRING cmd... -- rocprofv3 passes over any command, run from the repo root on the GPU box:
#   pass 0: --kernel-trace --stats (per-kernel times)            -> gpurun_out/OUT/stats/
#   passes 1..4: --pmc, one counter group per run (never mixed with a trace domain) -> gpurun_out/OUT/g<i>/
# Writes gpurun_out/OUT/summary.csv: counter, dispatches seen, per-dispatch mean of the kernels whose name contains
# KERNEL_SUBSTRING (the counter rows of one dispatch are summed over XCD / SE instances first), and
# gpurun_out/OUT/kernel_stats.csv (the --stats table).
set -u
OUT=gpurun_out/$1; shift
KSUB=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- "$@" > $R/$OUT/stats.log 2>&1
GROUPS_=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE")
i=0
for g in "${GROUPS_[@]}"; do
  timeout 600 rocprofv3 --pmc $g --output-format csv -d $R/$OUT/g$i -- "$@" > $R/$OUT/g$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections, shutil
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$R/$OUT/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "$KSUB" in row["Kernel_Name"]:
            per[row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
with open("$R/$OUT/summary.csv", "w") as o:
    o.write("counter,dispatches,per_dispatch_mean\n")
    for k in sorted(per):
        v = list(per[k].values())
        o.write("%s,%d,%e\n" % (k, len(v), sum(v) / len(v)))
for f in glob.glob("$R/$OUT/stats/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, "$R/$OUT/kernel_stats.csv")
print(open("$R/$OUT/summary.csv").read())
try:
    print(open("$R/$OUT/kernel_stats.csv").read()[:3000])
except OSError:
    pass
PY
