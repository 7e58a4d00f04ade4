#!/bin/bash
# Interleaved A/B of one environment switch on one box: $1 = "VAR=value" for the A leg (B = product default), rest -> bench.py
sw=$1; shift
for r in 1 2 3; do
  for leg in "$sw" "EDV_NOP=1"; do
    env "$leg" python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; a=d['roofline_attention']
print('$leg', d['value'], d['ms_per_step'], 'gemm', r['achieved'], 'enc', r['encoder_launches']['achieved'], 'attn', a['achieved'], 'hbm ms', d['roofline_hbm']['ms_per_step'])"
  done
done
