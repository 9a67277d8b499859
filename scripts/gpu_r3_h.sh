#!/bin/bash
# round 3, GPU call H: trained-scale tests, gradient tests (bucketed gradients), 1-rank RCCL test, training-step kernel profile
mkdir -p gpurun_out/r3h
timeout 1500 python -m pytest tests/test_gpu_trained.py tests/test_gpu_grad.py tests/test_gpu_infer.py -x -q -m gpu -s > gpurun_out/r3h/pytest.txt 2>&1
grep -n "passed\|failed\|sigma\|geometry\|Error" gpurun_out/r3h/pytest.txt | cut -c1-700
STEPS=4 bash scripts/prof_train.sh
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_train/kernel_stats.csv')))
for r in rows[:28]: print(r['Name'][:70], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2), r['Percentage'])
PY
