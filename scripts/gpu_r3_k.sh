#!/bin/bash
mkdir -p gpurun_out/r3u
timeout 1500 python -m pytest tests/test_gpu_grad.py tests/test_gpu_trained.py -x -q -m gpu > gpurun_out/r3u/pytest.txt 2>&1
tail -3 gpurun_out/r3u/pytest.txt | cut -c1-400
