#!/bin/bash
# rocprofv3 counter passes over the encode kernel only (tools/encode_only.py).  tools/pmc_encode.sh <outdir> [rays]
out=${1:-gpurun_out/pmc_enc}; rays=${2:-4096}
mkdir -p /root/repo/$out
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU"
 "SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_WAVES"
 "TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"
)
i=0
for g in "${groups[@]}"; do
  rm -rf /tmp/pe_$i
  timeout 240 rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pe_$i -- python3 /root/repo/tools/encode_only.py $rays 6 > /dev/null 2>/tmp/pe_$i.err
  python3 /root/repo/tools/summarize_rocprof.py /tmp/pe_$i 2>/dev/null | grep "k_encode_lane" > /root/repo/$out/group_$i.txt
  i=$((i+1))
done
