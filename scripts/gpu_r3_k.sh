#!/bin/bash
# round 3, GPU call: whole GPU suite + E inverse breakdown
mkdir -p gpurun_out/r3l
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3l/pytest.txt 2>&1
tail -6 gpurun_out/r3l/pytest.txt | cut -c1-300
python bench.py --config E --mode inverse --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('E inverse', d['value'], d['ms_per_step'], d['kernel_launches_per_step']); print(d['breakdown_ms_per_step'])"
