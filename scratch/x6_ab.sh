#!/bin/bash
# A/B of the products modes on the default bench (run on the GPU box from the repo root): value / serial value per setting
cd ${GRAFT_REPO_ROOT:-$PWD}
ARGS="$*"
for cfg in "f32 0" "bf16x6 0" "bf16x6 1" "bf16x6 2" "f32 1"; do
  set -- $cfg
  for fl in 1 0; do
    extra=""; [ $fl = 1 ] && extra="--in-flight 1"
    EDV_PRODUCTS=$1 EDV_ENC_STREAMS=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events $extra $ARGS > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; continue; }
    python - <<PY
import json
j=json.load(open("/tmp/ab.json"))
print("products=$1 enc_streams=$2 in_flight=%s value=%.1f ms=%.3f" % (j["config"].get("clips_in_flight"), j["value"], j["ms_per_step"]), flush=True)
PY
  done
done
