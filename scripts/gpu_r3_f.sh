#!/bin/bash
# round 3, GPU call F: whole GPU suite (init pass on the MFMA kernels, graph, secondary bench line), bench
mkdir -p gpurun_out/r3f
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3f/pytest.txt 2>&1
tail -15 gpurun_out/r3f/pytest.txt
(time python bench.py) > gpurun_out/r3f/bench.txt 2>&1
tail -4 gpurun_out/r3f/bench.txt | cut -c1-1500
