#!/bin/bash
# round 3, GPU call: finishing-kernel workgroup sizes A/B
mkdir -p gpurun_out/r3p
bash scripts/ab2.sh base default > gpurun_out/r3p/ab.txt 2>&1
cat gpurun_out/r3p/ab.txt | cut -c1-400
