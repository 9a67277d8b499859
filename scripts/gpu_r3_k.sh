#!/bin/bash
mkdir -p gpurun_out/r3t
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3t/pytest.txt 2>&1
tail -3 gpurun_out/r3t/pytest.txt | cut -c1-300
bash scripts/prof_all.sh r03
