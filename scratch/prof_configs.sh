#!/bin/bash
# bench lines + rocprofv3 kernel stats of BASELINE configs 2..5 (run from the repo root on the GPU box)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-r02u}
mkdir -p $O
cd $R
run() { # name, bench args...
  name=$1; shift
  timeout -k 10 400 python bench.py "$@" > $O/$name.json 2> $O/$name.err || echo "bench $name failed" >> $O/fail.log
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o k -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > $O/prof_$name.log 2>&1) || echo "rocprof $name failed" >> $O/fail.log
  f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${name}_kernel_stats.csv
  rm -rf $O/prof_$name
  echo "done $name: $(head -c 200 $O/$name.json)"
}
if [ "${2:-all}" = "all" ]; then
run c3_vitb_T16 --encoder vitb --T 16 --no-cpu-baseline --steps 10
run c5_vitl_T32 --encoder vitl --T 32 --no-cpu-baseline --steps 5 --warmup 2
run c4_train_vitb_T16_224x280 --train --encoder vitb --T 16 --image 224x280 --steps 10
run c4_train_vitb_T16_224x280_l1 --train --l1-loss --encoder vitb --T 16 --image 224x280 --steps 10
run c4_train_vitb_T16_518 --train --encoder vitb --T 16 --steps 5 --warmup 2
run train_vits_T8 --train --steps 10
run train_vits_T8_l1 --train --l1-loss --steps 10
fi
run c2_vits_T8 --no-cpu-baseline
