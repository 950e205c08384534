#!/bin/bash
# tools/pmc_mea.sh -- rocprofv3 PMC passes (one counter group per run) over the default bench (one step); per-kernel
# per-launch sums for the MEA kernels of npr_batch_finish -> gpurun_out/pmc_mea/summary.csv
set -u
OUT=gpurun_out/pmc_mea
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
GROUPS_=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE")
i=0
for g in "${GROUPS_[@]}"; do
  timeout 400 rocprofv3 --pmc $g --output-format csv -d $R/$OUT/g$i -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/$OUT/g$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in glob.glob("$R/$OUT/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_mea_" not in k:
            continue
        k = k.split("k_mea_")[1].split("(")[0]
        acc["k_mea_" + k][row["Counter_Name"]] += float(row["Counter_Value"])
        launches["k_mea_" + k].add((f, row["Dispatch_Id"]))
with open("$R/$OUT/summary.csv", "w") as o:
    o.write("kernel,counter,per_launch_sum\n")
    for k in sorted(acc):
        for c in sorted(acc[k]):
            n = max(1, len({d for f, d in launches[k] if True}) )
            o.write("%s,%s,%e\n" % (k, c, acc[k][c]))
print(open("$R/$OUT/summary.csv").read())
PY
