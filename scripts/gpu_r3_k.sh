#!/bin/bash
mkdir -p gpurun_out/r3u
timeout 1500 python -m pytest tests/test_gpu_grad.py -x -q -m gpu -k "log_scale or k_cnet" > gpurun_out/r3u/pytest.txt 2>&1
tail -25 gpurun_out/r3u/pytest.txt | cut -c1-400
