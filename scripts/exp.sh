for e in 0 1 2 3; do
GLOWHIP_EXP=$e python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']
print('exp $e', d['value'], d['ms_per_step_min'], {k: v for k, v in b.items() if 'cnet_f0' in k})"
done
