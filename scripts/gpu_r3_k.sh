#!/bin/bash
# round 3, GPU call: side-stream pack A/B (flag 0x20000000 = one stream), whole GPU suite
mkdir -p gpurun_out/r3n
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3n/pytest.txt 2>&1
tail -4 gpurun_out/r3n/pytest.txt | cut -c1-300
for round in 1 2; do for fl in 0x20000000 0; do
for cfg in B E; do
GLOWHIP_DEBUG_FLAGS=$fl python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-secondary --no-exact-leg 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$fl', '$cfg', d['value'], d['ms_per_step_min'])"
done; done; done
python bench.py --no-cpu-baseline --no-secondary --no-exact-leg 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('graph', d['launch'][:30], d['value'], d['ms_per_step_min'])"
