#!/bin/bash
# round 3, GPU call: k_first_sh for Cin up to 128: parity tests, E numbers
mkdir -p gpurun_out/r3r
timeout 2400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3r/pytest.txt 2>&1
tail -4 gpurun_out/r3r/pytest.txt | cut -c1-300
python bench.py --config E --steps 10 --warmup 3 --no-cpu-baseline --no-graph 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('E fwd', d['value'], d['ms_per_step_min'], d['kernel_launches_per_step']); print({k:v for k,v in d['breakdown_ms_per_step'].items() if 'C192' in k or 'C384' in k})"
python bench.py --config E --mode inverse --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('E inv', d['value'], d['ms_per_step_min'])"
