#!/bin/bash
# round 3, GPU call: kernel-variant A/B (libglowhip_base.so vs libglowhip.so) + parity tests of the fused path
mkdir -p gpurun_out/r3k
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3k/pytest.txt 2>&1
tail -3 gpurun_out/r3k/pytest.txt | cut -c1-300
bash scripts/ab2.sh base default > gpurun_out/r3k/ab.txt 2>&1
cat gpurun_out/r3k/ab.txt | cut -c1-400
