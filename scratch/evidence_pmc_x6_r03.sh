#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
D=${1:-r3pmcx}
mkdir -p gpurun_out/$D
export EVIDENCE_ONLY=pmc
E=scratch/evidence_r03.sh
$E c2_vits_T8_bf16x6 $D --steps 20 --warmup 5 --products bf16x6
$E c3_vitb_T16_bf16x6 $D --encoder vitb --T 16 --steps 8 --warmup 2 --products bf16x6
$E c5_vitl_T32_bf16x6 $D --encoder vitl --T 32 --steps 4 --warmup 1 --products bf16x6
