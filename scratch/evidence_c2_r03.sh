#!/bin/bash
# the headline configuration alone, both products modes (after a late kernel change): scratch/evidence_c2_r03.sh <outdir>
cd ${GRAFT_REPO_ROOT:-$PWD}
D=${1:-r3ev5}
mkdir -p gpurun_out/$D
E=scratch/evidence_r03.sh
$E c2_vits_T8_bf16x6 $D --steps 20 --warmup 5 --products bf16x6
$E c2_vits_T8_f32 $D --steps 20 --warmup 5 --products f32 --no-cpu-baseline
