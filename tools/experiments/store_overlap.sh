#!/bin/bash
# Round 5, review item 1: where do the training forward / backward lose their store overlap?
# Variants are built by tools/experiments/store_overlap_build.sh (patch store_overlap.patch on a scratch tree):
#   xbase  = shipped code + the runtime store predicate the patch adds        xnost = stores predicated off (VALU kept)
#   xscr   = stash / mask / dz stores aimed at a 4 KB-per-wave scratch (L2-resident), plain     xscrnt = the same, nt
#   xplain / xsc1 / xsc01 / xntsc1 = the real destination with plain / sc1 / sc0 sc1 / sc1 nt stores
# usage (GPU box): tools/experiments/store_overlap.sh <outdir> [rounds] [variants...]
out=${1:-gpurun_out/store_overlap}; rounds=${2:-3}; shift 2
vars=${@:-main xbase xnost xscr xscrnt xplain xsc1 xsc01 xntsc1}
mkdir -p $out
log=$out/timing.txt; : > $log
for rep in $(seq $rounds); do
  for v in $vars; do
    if [ "$v" = main ]; then unset DURF_LIB_PATH; else export DURF_LIB_PATH=durf_amd/variants/libdurf_$v.so; fi
    echo -n "variant=$v  " >> $log; python3 tools/time_fwd.py 2>&1 | tail -1 >> $log
  done
done
unset DURF_LIB_PATH
# counters: one group per pass, --pmc only
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
 "SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_ANY"
 "WRITE_SIZE TCC_EA0_WRREQ_STALL_sum"
 "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
)
for v in ${PMC_VARIANTS:-xbase xnost xscr}; do
  export DURF_LIB_PATH=$root/durf_amd/variants/libdurf_$v.so
  i=0
  for g in "${groups[@]}"; do
    rm -rf /tmp/so_$v_$i
    rocprofv3 --pmc $g --output-format csv -d /tmp/so_${v}_$i -- python3 $root/tools/experiments/bench_fwd_only.py > /dev/null 2>/tmp/so_${v}_$i.err
    python3 $root/tools/summarize_rocprof.py /tmp/so_${v}_$i > $root/$out/pmc_${v}_$i.txt 2>&1
    tail -2 /tmp/so_${v}_$i.err >> $root/$out/pmc_${v}_$i.txt
    i=$((i+1))
  done
done
