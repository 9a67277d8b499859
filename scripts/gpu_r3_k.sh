#!/bin/bash
mkdir -p gpurun_out/r3u
for v in 0x80000000 0 0x80000000 0; do
  GLOWHIP_DEBUG_FLAGS=$v python bench.py --mode train --steps 10 --warmup 4 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step_min'], d['config'].get('loss_mean_nll_bits_per_dim'))"
done
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3u/pytest.txt 2>&1
tail -5 gpurun_out/r3u/pytest.txt | cut -c1-400
