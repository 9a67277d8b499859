#!/bin/bash
mkdir -p gpurun_out/r3u
timeout 1500 python -m pytest tests/test_gpu_grad.py tests/test_gpu_trained.py -x -q -m gpu > gpurun_out/r3u/pytest.txt 2>&1
tail -3 gpurun_out/r3u/pytest.txt | cut -c1-400
for v in 0x4000000 0 0x4000000 0; do
  GLOWHIP_DEBUG_FLAGS=$v python bench.py --mode train --steps 10 --warmup 4 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('train $v', d['value'], d['ms_per_step_min'])"
done
