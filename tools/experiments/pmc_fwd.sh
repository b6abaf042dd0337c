#!/bin/bash
# rocprofv3 PMC passes over the fused-MLP microbench (tools/bench_fwd_only.py: inference fwd, training fwd, bwd):
# where do the waves' cycles go?   tools/pmc_fwd.sh <outdir>   (run on the GPU box)
out=${1:-gpurun_out/pmc_fwd}
mkdir -p /root/repo/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQC?_[A-Z0-9_]+)\b" | sort -u > /root/repo/$out/sq_counters.txt
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
 "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"
)
i=0
for g in "${groups[@]}"; do
  rm -rf /tmp/pmcf_$i
  rocprofv3 --pmc $g --output-format csv -d /tmp/pmcf_$i -- python3 /root/repo/tools/bench_fwd_only.py > /dev/null 2>/tmp/pmcf_$i.err
  python3 /root/repo/tools/summarize_rocprof.py /tmp/pmcf_$i > /root/repo/$out/pmc_$i.txt 2>&1
  tail -3 /tmp/pmcf_$i.err >> /root/repo/$out/pmc_$i.txt
  i=$((i+1))
done
