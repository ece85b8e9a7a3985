#!/bin/bash
# GPU box: what bounds kr_kb_tails (counter passes, development)
out=gpurun_out/pmc_tails
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_PERF_SEL_TOTAL_MISS_LRU_READ TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "kr_kb_tails" --output-format csv -d $out/p$i -- python bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline > $out/p$i.log 2>&1
  tail -2 $out/p$i.log
done
ls $out/*/*
