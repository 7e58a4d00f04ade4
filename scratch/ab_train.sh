#!/bin/bash
# Interleaved A/B of two builds on one box, fine-tune step: $1 = other .so, rest -> bench.py --train
other=$1; shift
for r in 1 2 3; do
  for lib in "$other" ""; do
    EDV_LIB_PATH=$lib python bench.py --train --no-kernel-events "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-product}'.split('/')[-1], d['value'], d['ms_per_step'])"
  done
done
