#!/bin/bash
# round 3, GPU call K: P2 B-prefetch A/B; C = 96 on k_cnet (parity tests + D / E secondary numbers)
mkdir -p gpurun_out/r3k
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3k/pytest.txt 2>&1
tail -5 gpurun_out/r3k/pytest.txt | cut -c1-300
bash scripts/ab2.sh base default > gpurun_out/r3k/ab.txt 2>&1
cat gpurun_out/r3k/ab.txt | cut -c1-400
python bench.py --config D --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('D', d['value'], d['ms_per_step'], d['kernel_launches_per_step'])"
python bench.py --config E --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('E', d['value'], d['ms_per_step'], d['kernel_launches_per_step']); print(d['breakdown_ms_per_step'])"
