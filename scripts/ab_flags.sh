#!/bin/bash
# A/B debug-flag variants of the default library: scripts/ab_flags.sh <hex flag>...   prints ms/step + kernel breakdown
for v in "$@"; do
  python bench.py --debug-flags $v --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']
print('$v', d['value'], d['ms_per_step'], 'f2', [b[k] for k in sorted(b) if 'f2' in k], 'f0', [b[k] for k in sorted(b) if 'f0' in k], 'f4', [b[k] for k in sorted(b) if 'f4' in k], 'mix', [b[k] for k in sorted(b) if 'chanmix' in k])"
done
