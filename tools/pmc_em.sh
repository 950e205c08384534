#!/bin/bash
# rocprofv3 PMC passes over the register E-step (tools/gpu_em_time.py n L W); prints per-launch means of k_em_stair
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_em
for g in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $g --output-format csv -d $R/gpurun_out/pmc_em/g_$RANDOM -- python $R/tools/gpu_em_time.py "$@" > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmc_em/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_em_stair" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc): print(k, len(acc[k]), sum(acc[k]) / 2.0)
PY
