#!/bin/bash
# GPU box: what bounds one kernel -- SQ / LDS / TA counter passes (one set per pass) on bench.py --config C3 --steps 1.
# usage: tools/pmc_kernel.sh <kernel regex> <tag>     (outputs under gpurun_out/<tag>/; reduce with tools/pmc_reduce.py)
re=$1
out=gpurun_out/$2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_TA_BUSY_sum TCP_LFIFO_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "$re" --output-format csv -d $out/p$i -- python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline > $out/p$i.log 2>&1
  tail -1 $out/p$i.log | cut -c1-200
done
