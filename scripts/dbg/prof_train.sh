cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r05
name=train_B
mkdir -p $O/$name
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o t -- python3 $R/bench.py --no-cpu-baseline --no-secondary --no-exact-leg --mode train --steps 5 --warmup 5 > $O/$name/log.txt 2>&1
grep '^{' $O/$name/log.txt | tail -1 > $O/$name/bench_line.json
python3 $R/scripts/trace_stats.py $(find $O/$name -name "*kernel_trace.csv" | head -1) $O/$name/steady_kernel_stats.csv
find $O/$name -name "*kernel_trace.csv" -delete
cut -c1-300 $O/$name/bench_line.json
