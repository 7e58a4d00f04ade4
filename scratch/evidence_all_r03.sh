#!/bin/bash
# Round-3 evidence from ONE build, one GPU box (run from the repo root):  scratch/evidence_all_r03.sh <outdir under gpurun_out>
# Every configuration goes through scratch/evidence_r03.sh: bench line, rocprofv3 --kernel-trace --stats, FETCH_SIZE / WRITE_SIZE passes, MFMA-busy pass.
# Inference configurations in both products modes (bf16x6 = the default, f32 = --products f32); the profiling passes run one arithmetic per run.
cd ${GRAFT_REPO_ROOT:-$PWD}
D=${1:-r3ev}
PART=${2:-all}   # "a": configs 2 and 3 + the ViT-S fine-tune step; "b": config 5 + config 4; "all"
mkdir -p gpurun_out/$D
E=scratch/evidence_r03.sh
[ "$PART" = "b" ] || $E c2_vits_T8_bf16x6 $D --steps 20 --warmup 5 --products bf16x6 && echo "c2 x6 ok" >> gpurun_out/$D/progress
[ "$PART" = "b" ] || $E c2_vits_T8_f32 $D --steps 20 --warmup 5 --products f32 --no-cpu-baseline && echo "c2 f32 ok" >> gpurun_out/$D/progress
[ "$PART" = "b" ] || $E c3_vitb_T16_bf16x6 $D --encoder vitb --T 16 --steps 8 --warmup 2 --products bf16x6 --no-cpu-baseline && echo "c3 x6 ok" >> gpurun_out/$D/progress
[ "$PART" = "b" ] || $E c3_vitb_T16_f32 $D --encoder vitb --T 16 --steps 8 --warmup 2 --products f32 --no-cpu-baseline && echo "c3 f32 ok" >> gpurun_out/$D/progress
[ "$PART" = "a" ] || $E c5_vitl_T32_bf16x6 $D --encoder vitl --T 32 --steps 4 --warmup 1 --products bf16x6 --no-cpu-baseline && echo "c5 x6 ok" >> gpurun_out/$D/progress
[ "$PART" = "a" ] || $E c5_vitl_T32_f32 $D --encoder vitl --T 32 --steps 4 --warmup 1 --products f32 --no-cpu-baseline && echo "c5 f32 ok" >> gpurun_out/$D/progress
[ "$PART" = "b" ] || $E train_vits_T8 $D --train --steps 10 --warmup 3 --no-cpu-baseline && echo "train ok" >> gpurun_out/$D/progress
[ "$PART" = "a" ] || $E c4_train_vitb_T16_224x280 $D --train --encoder vitb --T 16 --image 224x280 --steps 10 --warmup 3 --no-cpu-baseline && echo "c4 ok" >> gpurun_out/$D/progress
ls gpurun_out/$D | wc -l
