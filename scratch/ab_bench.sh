#!/bin/bash
# Interleaved A/B of two builds of the library on one box: $1 = the other .so (EDV_LIB_PATH), remaining args go to bench.py
other=$1; shift
for r in 1 2 3; do
  for lib in "$other" ""; do
    EDV_LIB_PATH=$lib python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; a=d['roofline_attention']
print('${lib:-product}'.split('/')[-1], d['value'], d['ms_per_step'], 'gemm', r['achieved'], 'enc', r['encoder_launches']['achieved'], 'attn', a['achieved'], 'hbm ms', d['roofline_hbm']['ms_per_step'])"
  done
done
