#!/bin/bash
# GPU box: hardware-counter passes over the KR SpMV kernel at C3 (one rocprofv3 --pmc run per counter group, each under
# its own timeout: a counter group the profiler cannot schedule otherwise hangs the whole call).
# usage: tools/pmc_kr.sh <tag> [env assignments for bench.py ...]    (outputs under gpurun_out/<tag>/pmc_<group>/)
tag=$1; shift
for kv in "$@"; do export "$kv"; done
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $group --kernel-include-regex "${B3C_PMC_KERNEL:-kr_kb_}" --output-format csv -d $out/pmc_$i -- python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline --no-variant-leg > $out/pmc_$i.log 2>&1
  echo "group $i: $group -> rc $?"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
GROUPS
python tools/pmc_sum.py $out
