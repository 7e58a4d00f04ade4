#!/bin/bash
# kernel-trace stats of the default bench in one products mode: scratch/x6_prof.sh <f32|bf16x6> <out name>
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp
export EDV_PRODUCTS=$1
O=$R/gpurun_out/$2; mkdir -p $O
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --in-flight 1 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
rm -rf $O/prof
head -12 $O/kernel_stats.csv | cut -c1-150
