#!/bin/bash
# Round profile set (run on the GPU box via gpurun; summaries are copied into profiles/ by the caller):
#   kernel trace + stats of the default bench command, of configs D / E, of the training step; HBM traffic PMC passes; SQ/TCC passes
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
run_trace() {  # name, bench args...
  local name=$1; shift
  mkdir -p $O/$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o t -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-exact-leg "$@" > $O/$name/log.txt 2>&1
  grep '^{' $O/$name/log.txt | tail -1 > $O/$name/bench_line.json
  # steady state only: the table without the one-off init pass (scripts/trace_stats.py)
  python3 $R/scripts/trace_stats.py $(find $O/$name -name "*kernel_trace.csv" | head -1) $O/$name/steady_kernel_stats.csv
  find $O/$name -name "*kernel_trace.csv" -delete      # large; the summaries are what gets committed
  echo "$name: $(cut -c1-160 $O/$name/bench_line.json)"
}
run_trace fwd_B --steps 10 --warmup 3
run_trace fwd_D --config D --steps 5 --warmup 2
run_trace fwd_E --config E --steps 5 --warmup 2
run_trace inv_E --config E --mode inverse --steps 5 --warmup 2
run_trace train_B --mode train --steps 5 --warmup 5      # (warm-up past TrainLoop.GRAPH_AFTER: the capture of the graphed step stays outside the timed steps)
bash $R/scripts/prof_pmc.sh > $O/pmc_log.txt 2>&1
bash $R/scripts/prof_sq.sh sq_$TAG > $O/sq_log.txt 2>&1
python3 $R/bench.py > $O/bench_B.json 2> $O/bench_B.err
tail -c 400 $O/bench_B.json
