#!/bin/bash
# tools/pmc_passes.sh OUT n L W -- rocprofv3 PMC passes (one counter group per run) over tools/gpu_pmc_run.py n L W
# (W = 0: the reference's anchor band).  Writes gpurun_out/pmc_OUT/summary.csv: counter, launches, per-launch mean of the
# kernels whose name contains $KERNEL (default k_dp_).
set -u
OUT=gpurun_out/pmc_$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
GROUPS_=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE")
i=0
for g in "${GROUPS_[@]}"; do
  timeout 300 rocprofv3 --pmc $g --output-format csv -d $R/$OUT/g$i -- python $R/tools/gpu_pmc_run.py "$@" > $R/$OUT/g$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$R/$OUT/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "${KERNEL:-k_dp_}" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("$R/$OUT/summary.csv", "w") as o:
    o.write("counter,launches,per_launch_mean\n")
    for k in sorted(acc):
        # one row per (dispatch, counter) -- possibly per XCD/SE instance: sum per dispatch
        o.write("%s,%d,%e\n" % (k, len(acc[k]), sum(acc[k]) / 2.0))
print(open("$R/$OUT/summary.csv").read())
PY
