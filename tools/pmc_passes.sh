#!/bin/bash
# rocprofv3 PMC passes (one counter group per run, --pmc only) over the training-kernel microbench.
#   tools/pmc_passes.sh <outdir>     (run on the GPU box; summaries land in <outdir>/pmc_<group>.txt)
out=${1:-gpurun_out/pmc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES"
 "SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
 "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum"
 "TCP_PENDING_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum"
)
i=0
for g in "${groups[@]}"; do
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $g --output-format csv -d /tmp/pmc_$i -- python3 /root/repo/tools/bench_mlp_train.py > /dev/null 2>/tmp/pmc_$i.err
  python3 /root/repo/tools/summarize_rocprof.py /tmp/pmc_$i > /root/repo/$out/pmc_$i.txt 2>&1
  i=$((i+1))
done
