#!/bin/bash
# Run on the GPU box: write-path counters of one conv_s3_kernel layer (scripts/conv_probe.py arguments after the tag)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_WRITE_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $R/scripts/conv_probe.py "$@" > $OUT/p$i.log 2>&1
  python $R/scripts/pmc_table.py $(find $OUT/p$i -name "*counter_collection.csv" | head -1) conv_s3_kernel
done
