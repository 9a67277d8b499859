#!/bin/bash
# round 3, GPU call: wide mixer + 256-pixel finishing workgroups: parity tests, D / E numbers
mkdir -p gpurun_out/r3o
timeout 2400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3o/pytest.txt 2>&1
tail -4 gpurun_out/r3o/pytest.txt | cut -c1-300
for cfg in D E; do
python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-graph 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg fwd', d['value'], d['ms_per_step_min']); print({k:v for k,v in d['breakdown_ms_per_step'].items() if 'finish' in k or 'chanmix' in k})"
done
python bench.py --config E --mode inverse --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('E inv', d['value'], d['ms_per_step_min']); print({k:v for k,v in d['breakdown_ms_per_step'].items() if 'finish' in k or 'chanmix' in k})"
