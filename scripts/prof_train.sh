#!/bin/bash
# Kernel-time breakdown of the training step (fwd with tape + backward) under rocprofv3; run on the GPU box via gpurun.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof_train
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -o train -- python3 $R/bench.py --mode train --steps ${STEPS:-4} --warmup 2 > $R/gpurun_out/prof_train/log.txt 2>&1
tail -2 $R/gpurun_out/prof_train/log.txt
find $R/gpurun_out/prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/prof_train/kernel_stats.csv
