cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace_gaps; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/log.txt 2>&1
python3 $R/scripts/trace_gaps.py $(find $O -name "*kernel_trace.csv" | head -1)
find $O -name "*kernel_trace.csv" -delete
