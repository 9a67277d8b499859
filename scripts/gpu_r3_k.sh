#!/bin/bash
# round 3, GPU call: whole GPU suite, then the default bench line (init times, secondary workloads)
mkdir -p gpurun_out/r3m
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3m/pytest.txt 2>&1
tail -4 gpurun_out/r3m/pytest.txt | cut -c1-300
python bench.py --no-cpu-baseline > gpurun_out/r3m/bench.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r3m/bench.txt'):
    if l.startswith('{'):
        d=json.loads(l)
        print(d['value'], d['ms_per_step'], 'init', d['data_dependent_init_ms'], 'enqueue', d['host_enqueue_ms_per_step'], d['roofline']['frac'])
        for k,v in d['secondary'].items(): print(k, v['value'], v['ms_per_step'], 'init', v.get('data_dependent_init_ms'), v['wall_s_incl_setup'])
PY
