#!/bin/bash
# A/B of library builds on ONE box, training step: scripts/ab_train2.sh <variant>...  (see ab2.sh for the variant names); two rounds each
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = "default" ]; then unset GLOWHIP_LIB_PATH; else export GLOWHIP_LIB_PATH=$PWD/pytorch-glow_amd/libglowhip_$v.so; fi
  python bench.py --mode train --steps 20 --warmup 10 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['ms_per_step_min'])"
done; done
