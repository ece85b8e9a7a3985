#!/bin/bash
# GPU box: kernel-trace stats + hardware-counter passes over the KR SpMV kernel (dictionary form) at C3 through the probe.
# usage: tools/pmc_tcd.sh <tag> [env assignments ...]
tag=$1; shift
for kv in "$@"; do export "$kv"; done
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python tools/tcd_probe.py C3 "" > $out/trace.log 2>&1
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $group --kernel-include-regex "${B3C_PMC_KERNEL:-kr_kb_tcd}" --output-format csv -d $out/pmc_$i -- python tools/tcd_probe.py C3 "" > $out/pmc_$i.log 2>&1
  echo "group $i: $group -> rc $?"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
FETCH_SIZE
GROUPS
python tools/pmc_sum.py $out
python tools/kstats.py $out/trace 2>/dev/null | head -30
