#!/bin/bash
# round 3, GPU call J: per-wave stamps of k_cnet, gradient / trained-scale tests, training bench + profile
mkdir -p gpurun_out/r3j
for L in 1 3; do echo "== L=$L"; K=1 L=$L python scripts/stamps_cnet_waves.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r3j/stamps_waves.txt 2>&1
cat gpurun_out/r3j/stamps_waves.txt
timeout 1500 python -m pytest tests/test_gpu_trained.py tests/test_gpu_grad.py tests/test_gpu_infer.py -x -q -m gpu -s > gpurun_out/r3j/pytest.txt 2>&1
grep -n "passed\|failed\|sigma\|geometry\|Error" gpurun_out/r3j/pytest.txt | cut -c1-900
python bench.py --mode train --steps 8 --warmup 3 2>&1 | tail -1 | cut -c100-330
STEPS=4 bash scripts/prof_train.sh
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_train/kernel_stats.csv')))
for r in rows[:16]: print(r['Name'][:70], r['Calls'], round(float(r['TotalDurationNs'])/1e6/6,2), r['Percentage'])
PY
