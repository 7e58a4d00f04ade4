#!/bin/bash
# Interleaved A/B/C... of environment settings on one box: each argument before "--" is one leg ("VAR=v VAR2=w" or "default"); the rest goes to bench.py
legs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do legs+=("$1"); shift; done
shift
for r in 1 2 3; do
  for leg in "${legs[@]}"; do
    if [ "$leg" = "default" ]; then envs=(EDV_NOP=1); else read -r -a envs <<< "$leg"; fi
    env "${envs[@]}" python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; a=d['roofline_attention']
print('%-36s' % '$leg', d['value'], d['ms_per_step'], 'gemm', r['achieved'], 'enc', r['encoder_launches']['achieved'], 'attn', a['achieved'], 'hbm ms', d['roofline_hbm']['ms_per_step'], 'launches', d['launches_per_step'])"
  done
done
