#!/bin/bash
# round 3, GPU call C: stamps of k_cnet at levels 1 / 2 / 3, range / family / optimiser tests, bench
mkdir -p gpurun_out/r3c
for L in 1 2 3; do echo "== L=$L"; K=1 L=$L python scripts/stamps_cnet.py; done > gpurun_out/r3c/stamps.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_grad.py tests/test_gpu_fused.py tests/test_gpu_infer.py -x -q -m gpu > gpurun_out/r3c/pytest.txt 2>&1
tail -15 gpurun_out/r3c/pytest.txt
python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/r3c/bench.txt 2>&1
cat gpurun_out/r3c/stamps.txt; tail -1 gpurun_out/r3c/bench.txt
