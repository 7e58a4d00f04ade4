#!/bin/bash
# end-to-end: fp32 vs bf16x6 (linears only) vs bf16x6 (linears + attention), configs 2 / 3 / 5; then the model parity tests in bf16x6 mode
cd ${GRAFT_REPO_ROOT:-$PWD}
for cfg in "vits 8 20 5" "vitb 16 8 2" "vitl 32 4 1"; do
  set -- $cfg
  for m in "f32 1" "bf16x6 0" "bf16x6 1"; do
    set -- $cfg $m
    EDV_X6_ATTN=$6 python bench.py --encoder $1 --T $2 --steps $3 --warmup $4 --products $5 --no-cpu-baseline --no-kernel-events --no-other-products > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; continue; }
    python - <<PY
import json
j=json.load(open("/tmp/ab.json"))
print("$1 T=$2 products=$5 x6_attn=$6 value=%.2f ms=%.3f serial=%s" % (j["value"], j["ms_per_step"], j.get("one_clip_at_a_time_value")), flush=True)
PY
  done
done
[ "$1" = "notest" ] || EDV_PRODUCTS=bf16x6 timeout -k 10 900 python -m pytest tests/test_forward_gpu.py tests/test_baseline_configs_gpu.py tests/test_video_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -5
