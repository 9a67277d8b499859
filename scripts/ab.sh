#!/bin/bash
# A/B kernel variants: scripts/ab.sh <lib-suffix>...  ('' = the default build); prints ms/step + kernel breakdown
for v in "$@"; do
  if [ "$v" = "base" ]; then unset GLOWHIP_LIB_PATH; else export GLOWHIP_LIB_PATH=$PWD/pytorch-glow_amd/libglowhip_$v.so; fi
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']
print('$v', d['value'], d['ms_per_step'], 'gemm TF', d['roofline']['achieved'], 'f2', [b[k] for k in sorted(b) if 'f2' in k], 'f0', [b[k] for k in sorted(b) if 'f0' in k], 'f4', [b[k] for k in sorted(b) if 'f4' in k])"
done
