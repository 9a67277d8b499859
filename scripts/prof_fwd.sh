#!/bin/bash
# Kernel trace of the forward bench step under rocprofv3 (run on the GPU box via gpurun): per-kernel stats + raw trace
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof_fwd
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fwd -o fwd -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_fwd/log.txt 2>&1
tail -1 $R/gpurun_out/prof_fwd/log.txt | cut -c1-300
