#!/bin/bash
# A/B of library builds on ONE box: scripts/ab2.sh <variant>...   ("default" = pytorch-glow_amd/libglowhip.so; other names:
# pytorch-glow_amd/libglowhip_<name>.so built with `make BUILD=build_<name> LIB=../libglowhip_<name>.so`); two rounds each
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = "default" ]; then unset GLOWHIP_LIB_PATH; else export GLOWHIP_LIB_PATH=$PWD/pytorch-glow_amd/libglowhip_$v.so; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']
print('$v', d['value'], d['ms_per_step_min'], {k: v for k, v in b.items() if 'cnet' in k})"
done; done
