#!/bin/bash
# SQ / TCC / GRBM counters of the forward bench step (run on the GPU box via gpurun).  One rocprofv3 --pmc pass per counter
# group (8 SQ slots, 4 TCC, 2 GRBM per pass), kernel trace only, program directly after `--`.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-sq}
mkdir -p $R/gpurun_out/$TAG
rocprofv3 -L > $R/gpurun_out/$TAG/counters_list.txt 2>&1
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  d=$R/gpurun_out/$TAG/p$i
  mkdir -p $d
  echo "$group" > $d/counters.txt
  timeout 300 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $d -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-exact-leg --no-graph ${BENCH_ARGS} > $d/log.txt 2>&1
  echo "pass $i ($group): rc=$? $(ls $d | tr '\n' ' ')"
done <<'EOG'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE GRBM_COUNT
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SALU
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
EOG
