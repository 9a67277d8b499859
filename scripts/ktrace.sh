#!/bin/bash
# Per-kernel durations of the bench step (run on the GPU box via gpurun): rocprofv3 kernel trace, top kernels by total time.
# usage: scripts/ktrace.sh <tag> [bench args...]   (kernel-variant switches: pass --debug-flags <value> among the bench args)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-kt}; shift
mkdir -p $R/gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o t -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/$TAG/log.txt 2>&1
tail -1 $R/gpurun_out/$TAG/log.txt | cut -c1-330
python3 - "$R/gpurun_out/$TAG/t_kernel_trace.csv" <<'PY'
import csv, collections, sys
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void glowhip::", "").replace("glowhip::", "")[-44:]
    if any(s in name for s in ("k_conv_direct", "k_actnorm_init", "rocclr", "at::")):
        continue
    k = (name, r.get("Grid_Size_X"), r.get("Grid_Size_Y"))
    agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{k[0]:46s} grid={k[1]:>7s},{k[2]:>2s} n={n:5d} avg_us={t / n / 1e3:9.2f} total_ms={t / 1e6:8.2f}")
PY
