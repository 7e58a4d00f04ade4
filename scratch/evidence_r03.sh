#!/bin/bash
# Round-3 evidence for one configuration, all from the same build (run from the repo root on the GPU box):
#   scratch/evidence_r03.sh <name> <outdir under gpurun_out> [bench args...]
# 1. the bench line                          -> <out>/<name>.json
# 2. rocprofv3 --kernel-trace --stats        -> <out>/<name>_kernel_stats.csv
# 3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; inference configs only) -> <out>/<name>_traffic.json   (per-launch bytes per kernel class)
# 4. one --pmc pass SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE      -> <out>/<name>_mfma_busy.txt   (vs SQ-busy and vs wall)
# Under rocprofv3 the program after "--" is python3 itself (no env / bash -c hop); bench runs one clip at a time there (--in-flight 1) so that a
# kernel's counters are its own.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
name=$1; O=$R/gpurun_out/$2; shift 2
ONLY=${EVIDENCE_ONLY:-all}   # "mfma": only the MFMA-busy pass; "pmc": the three counter passes, no bench line / kernel stats
mkdir -p $O
cd $R
if [ "$ONLY" = "all" ]; then
timeout -k 10 500 python bench.py "$@" > $O/$name.json 2> $O/$name.err || echo "bench $name failed" >> $O/fail.log
echo "bench $name: $(head -c 160 $O/$name.json)"
fi
PROF="--no-cpu-baseline --no-kernel-events --no-other-products --steps 4 --warmup 2"
case " $* " in *" --train "*) ;; *) PROF="$PROF --in-flight 1";; esac
[ "$ONLY" != "all" ] || (cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o k -- python3 $R/bench.py "$@" $PROF > $O/prof_$name.log 2>&1) || echo "rocprof $name failed" >> $O/fail.log
f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${name}_kernel_stats.csv
rm -rf $O/prof_$name
echo "stats $name done"
case " $* " in *" --train "*) exit 0;; esac
[ "$ONLY" != "mfma" ] && for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${name}_$c -o k -- python3 $R/bench.py "$@" $PROF > $O/pmc_${name}_$c.log 2>&1) || echo "pmc $c $name failed" >> $O/fail.log
done
(cd /tmp && timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_${name}_mfma -o k -- python3 $R/bench.py "$@" $PROF > $O/pmc_${name}_mfma.log 2>&1) || echo "pmc mfma $name failed" >> $O/fail.log
python3 scratch/evidence_parse_r03.py $O $name "$@"
rm -rf $O/pmc_${name}_FETCH_SIZE $O/pmc_${name}_WRITE_SIZE $O/pmc_${name}_mfma
echo "pmc $name done"
