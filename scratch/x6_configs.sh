#!/bin/bash
# end-to-end A/B of the products modes on configs 3 and 5, then the model parity tests with bf16x6 as the initial mode
cd ${GRAFT_REPO_ROOT:-$PWD}
for cfg in "vitb 16" "vitl 32"; do
  set -- $cfg
  for m in f32 bf16x6; do
    EDV_PRODUCTS=$m python bench.py --encoder $1 --T $2 --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-events > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; continue; }
    python - <<PY
import json
j=json.load(open("/tmp/ab.json"))
print("$1 T=$2 products=$m value=%.2f ms=%.3f in_flight=%s" % (j["value"], j["ms_per_step"], j["config"].get("clips_in_flight")), flush=True)
PY
  done
done
EDV_PRODUCTS=bf16x6 timeout -k 10 900 python -m pytest tests/test_forward_gpu.py tests/test_baseline_configs_gpu.py tests/test_video_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -8
