#!/bin/bash
mkdir -p gpurun_out/r3u
timeout 1500 python -m pytest tests/test_gpu_grad.py -x -q -m gpu -k "other_couplings" > gpurun_out/r3u/pytest.txt 2>&1
tail -30 gpurun_out/r3u/pytest.txt | cut -c1-300
