#!/bin/bash
# host facts of the GPU box (cores visible / usable), for DESIGN.md and the cpu_baseline
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python3 -c "import os;print('affinity',len(os.sched_getaffinity(0)),'cpu_count',os.cpu_count())"; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" ; free -g | head -2
