#!/bin/bash
# HBM traffic counters of the forward bench step (run on the GPU box via gpurun): separate --pmc passes, kernel trace only
# (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in KB per dispatch; FETCH_SIZE x2 on gfx950 for 16 B/lane coalesced reads).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $R/gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-exact-leg --no-graph > $R/gpurun_out/pmc_$c/log.txt 2>&1
  tail -1 $R/gpurun_out/pmc_$c/log.txt | cut -c1-200
  ls $R/gpurun_out/pmc_$c | head
done
