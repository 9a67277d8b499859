for f in 0 0x8000000; do
GLOWHIP_DEBUG_FLAGS=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']
print('flags $f', d['value'], d['ms_per_step_min'], {k: v for k, v in b.items() if 'cnet' in k})"
done
