#!/bin/bash
# A/B of kernel-variant switches on ONE box, training step of config B: scripts/ab_flags_train.sh <debug flag value>...  (two rounds)
for round in 1 2; do
for f in "$@"; do
  python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --debug-flags $f 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'])"
done; done
