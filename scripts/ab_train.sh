#!/bin/bash
# A/B of library builds on ONE box, training step of config B: scripts/ab_train.sh <variant>...  (names as in ab2.sh)
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = "default" ]; then unset GLOWHIP_LIB_PATH; else export GLOWHIP_LIB_PATH=$PWD/pytorch-glow_amd/libglowhip_$v.so; fi
  python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-secondary ${BENCH_ARGS} 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d.get('breakdown_ms_per_step') or {}
print('$v', d['value'], d['ms_per_step'], {k: v for k, v in b.items() if 'wgrad' in k})"
done; done
