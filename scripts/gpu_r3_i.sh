#!/bin/bash
# round 3, GPU call I: trained-scale tests, gradient tests, training-step A/B + profile
mkdir -p gpurun_out/r3i
timeout 1500 python -m pytest tests/test_gpu_trained.py tests/test_gpu_grad.py tests/test_gpu_infer.py -x -q -m gpu -s > gpurun_out/r3i/pytest.txt 2>&1
grep -n "passed\|failed\|sigma\|geometry\|Error" gpurun_out/r3i/pytest.txt | cut -c1-900
for v in base default base default; do
  if [ "$v" = "default" ]; then unset GLOWHIP_LIB_PATH; else export GLOWHIP_LIB_PATH=$PWD/pytorch-glow_amd/libglowhip_$v.so; fi
  python bench.py --mode train --steps 8 --warmup 3 2>&1 | tail -1 | cut -c100-330
done > gpurun_out/r3i/train_ab.txt 2>&1
cat gpurun_out/r3i/train_ab.txt
unset GLOWHIP_LIB_PATH
STEPS=4 bash scripts/prof_train.sh
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_train/kernel_stats.csv')))
for r in rows[:22]: print(r['Name'][:70], r['Calls'], round(float(r['TotalDurationNs'])/1e6/6,2), r['Percentage'])
PY
