#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
D=${1:-r3ev3}
mkdir -p gpurun_out/$D
E=scratch/evidence_r03.sh
$E c2_vits_T8 $D --steps 20 --warmup 5 && echo "c2 ok" >> gpurun_out/$D/progress
$E c2_vits_T8_bf16x6 $D --steps 20 --warmup 5 --products bf16x6 --no-cpu-baseline && echo "c2 x6 ok" >> gpurun_out/$D/progress
