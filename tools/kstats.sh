#!/bin/bash
# tools/kstats.sh OUT cmd... -- rocprofv3 --kernel-trace --stats of a command (run from the repo root on the GPU box); prints the
# per-kernel table and leaves it at gpurun_out/OUT/kernel_stats.csv.  Bring-up tool.
set -u
OUT=gpurun_out/$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- "$@" > $R/$OUT/stats.log 2>&1 < /dev/null
f=$(ls $R/$OUT/stats/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $R/$OUT/kernel_stats.csv; cut -d, -f1-4 "$f" | head -${KSTATS_TOP:-16}; else echo "no kernel_stats.csv"; tail -5 $R/$OUT/stats.log; fi
