#!/bin/bash
# kernel trace of the training step for each named library variant (as ab2.sh), per (kernel, grid) table: scripts/trace_train.sh <variant>... [-- pattern]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = "default" ]; then unset GLOWHIP_LIB_PATH; else export GLOWHIP_LIB_PATH=$R/pytorch-glow_amd/libglowhip_$v.so; fi
  O=$R/gpurun_out/trace_train_$v; rm -rf $O; mkdir -p $O
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > $O/log.txt 2>&1
  echo "== $v: $(grep '^{' $O/log.txt | tail -1 | cut -c1-150)"
  python3 $R/scripts/trace_by_grid.py $(find $O -name "*kernel_trace.csv" | head -1) "${PAT:-wgrad}" 0.5
  find $O -name "*kernel_trace.csv" -delete
done
