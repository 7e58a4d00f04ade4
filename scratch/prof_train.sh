#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-prof_train}; shift; mkdir -p $O
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o k -- python3 $R/bench.py --train --steps 5 --warmup 2 --no-kernel-events "$@" > $O/log.txt 2>&1)
f=$(find $O/p -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; rm -rf $O/p
