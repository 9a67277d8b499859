#!/bin/bash
# Copy the summaries of a scripts/prof_all.sh run (gpurun_out/prof_<tag>, gpurun_out/sq_<tag>, gpurun_out/pmc_*) into profiles/<tag>_*
# (run in the build container after the gpurun call): scripts/collect_profiles.sh r04
TAG=${1:-r04}
R=$(cd "$(dirname "$0")/.." && pwd)
for w in fwd_B fwd_D fwd_E inv_E train_B; do
  cp $R/gpurun_out/prof_$TAG/$w/steady_kernel_stats.csv $R/profiles/${TAG}_${w}_kernel_stats.csv
  python3 -c "import json,sys; d=json.load(open('$R/gpurun_out/prof_$TAG/$w/bench_line.json')); json.dump(d, open('$R/profiles/${TAG}_${w}_bench.json','w'), indent=1)"
done
python3 -c "import json; d=json.loads(open('$R/gpurun_out/prof_$TAG/bench_B.json').read().strip().splitlines()[-1]); json.dump(d, open('$R/profiles/${TAG}_bench.json','w'), indent=1)"
python3 $R/scripts/pmc_summary.py $TAG | tail -3
python3 $R/scripts/sq_summary.py sq_$TAG $TAG | tail -3
ls -la $R/profiles | grep $TAG
